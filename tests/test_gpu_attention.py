"""Paged verify/decode attention and rope+append against the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle.attention import unpack_pages, verify_attention
from tests.gpu_util import pack_cache, pack_cache_fast, rb

pytestmark = pytest.mark.gpu


def _ops():
    from vllm_omni_draft_amd import ops
    return ops


def _setup(B, ctx_lens, q_len, Hq, Hkv, D, seed, shuffle=True):
    """Random K/V for every sequence (context + the q_len new tokens), packed into a
    shared page pool with shuffled block tables."""
    g = torch.Generator().manual_seed(seed)
    max_len = max(ctx_lens) + q_len
    pages_per_seq = (max_len + 15) // 16
    num_pages = B * pages_per_seq + 3
    perm = torch.randperm(num_pages, generator=g) if shuffle else torch.arange(num_pages)
    bt = perm[:B * pages_per_seq].view(B, pages_per_seq).to(torch.int32)
    kc = torch.zeros(num_pages, Hkv, D * 16, dtype=torch.bfloat16)
    vc = torch.zeros_like(kc)
    ks, vs, qs = [], [], []
    for b in range(B):
        n = ctx_lens[b] + q_len
        k = torch.randn(n, Hkv, D, generator=g).to(torch.bfloat16)
        v = torch.randn(n, Hkv, D, generator=g).to(torch.bfloat16)
        if n > 600 or B > 4:
            pack_cache_fast(kc, vc, k, v, bt[b])
        else:
            kb, vb = pack_cache(k, v, bt[b], num_pages)
            used = bt[b].long()
            kc[used], vc[used] = kb[used], vb[used]
        ks.append(k); vs.append(v)
        qs.append(torch.randn(q_len, Hq, D, generator=g).to(torch.bfloat16))
    return kc, vc, bt, ks, vs, qs


CASES = [
    # B, ctx_lens, q_len, Hq, Hkv, D, split
    (1, [0], 1, 4, 2, 64, 128),
    (1, [15], 2, 4, 2, 64, 128),
    (1, [16], 6, 4, 2, 64, 128),
    (1, [17], 6, 32, 8, 128, 256),
    (1, [100], 6, 32, 8, 128, 256),
    (2, [300, 7], 6, 32, 8, 128, 128),       # several splits + merge, ragged batch
    (3, [700, 129, 31], 2, 32, 8, 64, 256),   # draft-style first pass (2 tokens)
    (1, [260], 6, 28, 4, 128, 128),           # group of 7 heads: two 32-row M tiles
    (1, [500], 8, 8, 2, 128, 256),            # 8-token chunk (prefill path)
    (2, [40, 1023], 4, 12, 12, 64, 256),      # MHA (OPT-like)
]


@pytest.mark.parametrize("B,ctx_lens,q_len,Hq,Hkv,D,split", CASES)
def test_paged_attention_matches_oracle(B, ctx_lens, q_len, Hq, Hkv, D, split):
    ops = _ops()
    kc, vc, bt, ks, vs, qs = _setup(B, ctx_lens, q_len, Hq, Hkv, D, seed=sum(ctx_lens) + q_len)
    q = torch.cat(qs).reshape(B * q_len, Hq * D).cuda()
    seq_lens = torch.tensor(ctx_lens, dtype=torch.int32).cuda()   # first new position = ctx
    max_ctx = max(ctx_lens) + q_len
    out = ops.paged_attention(q, kc.cuda(), vc.cuda(), bt.cuda(), seq_lens, 0, B, q_len, Hq, Hkv, D,
                              D ** -0.5, max_ctx, split)
    torch.cuda.synchronize()
    out = out.float().cpu().view(B, q_len, Hq, D)
    for b in range(B):
        pos = [ctx_lens[b] + t for t in range(q_len)]
        ref = verify_attention(qs[b], ks[b], vs[b], pos, D ** -0.5)
        torch.testing.assert_close(out[b], ref, rtol=2e-2, atol=2e-2)


def _arm(B, q_len, Hq, Hkv, D, max_ctx, split):
    """The paged_attn_kernel<D, MT, NBUF, BT_PER_THREAD, RELOAD> instantiation a call takes
    (the dispatch of launch_attn in csrc/sd_attention.hip, restated)."""
    MT = 1 if q_len * (Hq // Hkv) <= 32 else 2
    nbuf = {(64, 1): 4, (64, 2): 2, (128, 1): 3, (128, 2): 1}[(D, MT)]
    tpw = split // 128
    n_splits = max(1, -(-max_ctx // split))
    if tpw <= nbuf:
        return (D, MT, tpw, 1, False)
    if MT == 1 and B * Hkv * n_splits >= 512:
        return (D, MT, 1, 1 if split // 16 + 8 <= 256 else 4, True)
    return (D, MT, nbuf, 1 if split // 16 + 8 * nbuf <= 256 else 4, True)


def _ragged(B, lo, hi, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(lo, hi, (B,), generator=g).tolist()


# Every dispatch arm of launch_attn against the oracle: ring (RELOAD) kernels at every depth,
# four block-table entries per thread (splits beyond ~3.5k keys), the one-tile ring with two
# workgroups per CU (>= 512 workgroups), lengths that end in the middle of a split and of a
# 32-key tile.  (The non-ring arms with splits of 128 / 256 keys are in CASES above.)
ARM_CASES = [
    # B, ctx_lens, q_len, Hq, Hkv, D, split
    (1, [1000], 6, 32, 8, 128, 384),                # <128,1,3,1,F>
    (2, [1500, 333], 6, 32, 8, 128, 512),           # <128,1,3,1,T>: ring of 3, several splits
    (2, [2300, 1029], 6, 32, 8, 128, 1024),         # <128,1,3,1,T>
    (2, [4100, 777], 6, 32, 8, 128, 4096),          # <128,1,3,4,T>: 4 table entries per thread
    (64, _ragged(64, 30, 700, 1) , 6, 16, 8, 128, 1024),   # <128,1,1,1,T>: 512 workgroups
    (64, _ragged(64, 30, 500, 2)[:63] + [4200], 6, 16, 8, 128, 4096),  # <128,1,1,4,T>: 1024 wgs
    (1, [1300], 6, 28, 4, 128, 512),                # <128,2,1,1,T>: two M tiles, one-tile ring
    (1, [4200], 6, 28, 4, 128, 4096),               # <128,2,1,4,T>
    (2, [900, 100], 2, 32, 8, 64, 384),             # <64,1,3,1,F>
    (2, [1100, 515], 2, 32, 8, 64, 512),            # <64,1,4,1,F>
    (2, [2100, 640], 2, 32, 8, 64, 1024),           # <64,1,4,1,T>
    (1, [4300], 1, 32, 8, 64, 4096),                # <64,1,4,4,T>
    (64, _ragged(64, 30, 600, 3), 2, 32, 8, 64, 1024),     # <64,1,1,1,T>
    (64, _ragged(64, 30, 400, 4)[:63] + [4150], 2, 32, 8, 64, 4096),   # <64,1,1,4,T>
    (1, [1000], 6, 14, 2, 64, 256),                 # <64,2,2,1,F>
    (1, [1200], 6, 14, 2, 64, 512),                 # <64,2,2,1,T>
    (1, [4200], 6, 14, 2, 64, 4096),                # <64,2,2,4,T>
]

ALL_ARMS = {
    (128, 1, 1, 1, False), (128, 1, 2, 1, False), (128, 1, 3, 1, False), (128, 1, 3, 1, True),
    (128, 1, 3, 4, True), (128, 1, 1, 1, True), (128, 1, 1, 4, True),
    (128, 2, 1, 1, False), (128, 2, 1, 1, True), (128, 2, 1, 4, True),
    (64, 1, 1, 1, False), (64, 1, 2, 1, False), (64, 1, 3, 1, False), (64, 1, 4, 1, False),
    (64, 1, 4, 1, True), (64, 1, 4, 4, True), (64, 1, 1, 1, True), (64, 1, 1, 4, True),
    (64, 2, 1, 1, False), (64, 2, 2, 1, False), (64, 2, 2, 1, True), (64, 2, 2, 4, True),
}


def _arm_of_case(c):
    B, ctx_lens, q_len, Hq, Hkv, D, split = c
    return _arm(B, q_len, Hq, Hkv, D, max(ctx_lens) + q_len, split)


def _arm_id(c):
    D, MT, nbuf, btp, reload_ = _arm_of_case(c)
    return f"attn<{D},{MT},{nbuf},{btp},{'ring' if reload_ else 'whole'}>-B{c[0]}-split{c[6]}"


def test_every_dispatch_arm_has_an_oracle_case():
    extra = [(1, [260], 6, 28, 4, 128, 128), (1, [300], 6, 14, 2, 64, 128)]   # two-tile, 128-key splits
    got = {_arm_of_case(c) for c in CASES + ARM_CASES + extra}
    assert got == ALL_ARMS, (ALL_ARMS - got, got - ALL_ARMS)


@pytest.mark.parametrize("case", ARM_CASES + [(1, [300], 6, 14, 2, 64, 128)], ids=_arm_id)
def test_paged_attention_arm_matches_oracle(case):
    B, ctx_lens, q_len, Hq, Hkv, D, split = case
    ops = _ops()
    kc, vc, bt, ks, vs, qs = _setup(B, ctx_lens, q_len, Hq, Hkv, D, seed=sum(ctx_lens) % 1000 + split)
    q = torch.cat(qs).reshape(B * q_len, Hq * D).cuda()
    seq_lens = torch.tensor(ctx_lens, dtype=torch.int32).cuda()
    max_ctx = max(ctx_lens) + q_len
    out = ops.paged_attention(q, kc.cuda(), vc.cuda(), bt.cuda(), seq_lens, 0, B, q_len, Hq, Hkv, D,
                              D ** -0.5, max_ctx, split)
    torch.cuda.synchronize()
    out = out.float().cpu().view(B, q_len, Hq, D)
    # the oracle on every sequence of a small batch, on a spread of sequences of a large one
    # (always the longest and the shortest)
    order = sorted(range(B), key=lambda b: ctx_lens[b])
    check = range(B) if B <= 4 else sorted({order[0], order[-1], order[B // 2], 0, B - 1, 17, 40})
    for b in check:
        pos = [ctx_lens[b] + t for t in range(q_len)]
        ref = verify_attention(qs[b], ks[b], vs[b], pos, D ** -0.5)
        torch.testing.assert_close(out[b], ref, rtol=2e-2, atol=2e-2)
    assert torch.isfinite(out).all()


@pytest.mark.parametrize("Hq,Hkv,D", [(32, 8, 128), (28, 4, 128), (8, 1, 128), (14, 2, 64)])
@pytest.mark.parametrize("split,max_pad", [(128, 0), (256, 0), (128, 1000)])
def test_verify_rows_equal_single_token_decode(split, max_pad, Hq, Hkv, D):
    """Bit-exact: token t scored inside a (k+1)-token verify pass == the same token
    scored alone at the same position, for any batch mix and any max_ctx bound.  Groups of 7
    and 8 heads put the verify pass on the two-M-tile kernel and the single token on the
    one-tile kernel: two instantiations that must give a row the same bits."""
    ops = _ops()
    B, ctx_lens, q_len = 2, [333, 18], 6
    kc, vc, bt, ks, vs, qs = _setup(B, ctx_lens, q_len, Hq, Hkv, D, seed=3)
    pad = (max_pad + 15) // 16
    bt = torch.cat([bt, torch.zeros(B, pad, dtype=torch.int32)], dim=1).contiguous()
    kc, vc, btc = kc.cuda(), vc.cuda(), bt.cuda()
    q = torch.cat(qs).reshape(B * q_len, Hq * D).cuda()
    seq_lens = torch.tensor(ctx_lens, dtype=torch.int32).cuda()
    max_ctx = max(ctx_lens) + q_len + max_pad
    full = ops.paged_attention(q, kc, vc, btc, seq_lens, 0, B, q_len, Hq, Hkv, D, D ** -0.5, max_ctx,
                               split).view(B, q_len, Hq * D)
    for b in range(B):
        for t in range(q_len):
            one = ops.paged_attention(q.view(B, q_len, -1)[b, t:t + 1].contiguous(), kc, vc,
                                      btc[b:b + 1].contiguous(), seq_lens[b:b + 1].contiguous(), t, 1, 1,
                                      Hq, Hkv, D, D ** -0.5, ctx_lens[b] + t + 1, split)
            assert torch.equal(one[0], full[b, t]), (b, t)


@pytest.mark.parametrize("Hq,Hkv,D,rope", [(4, 2, 64, True), (32, 8, 128, True), (12, 12, 64, False)])
def test_rope_append_layout_and_values(Hq, Hkv, D, rope):
    ops = _ops()
    from vllm_omni_draft_amd.config import create_model_config, rope_table
    g = torch.Generator().manual_seed(D + Hq)
    n, B = 45, 2
    cfg = create_model_config(None, head_dim=D, n_heads=Hq, n_kv_heads=Hkv, max_pos=64)
    cs = rope_table(cfg) if rope else None
    pages = 8
    bt = torch.tensor([[5, 2, 7], [1, 6, 0]], dtype=torch.int32)
    kc = torch.zeros(pages, Hkv, D * 16, dtype=torch.bfloat16).cuda()
    vc = torch.zeros_like(kc)
    qkv = torch.randn(B, n, (Hq + 2 * Hkv) * D, generator=g).to(torch.bfloat16)
    q_out = torch.zeros(B, n, Hq * D)
    seq0 = torch.tensor([0, 0], dtype=torch.int32).cuda()
    c0 = 0
    for q_len in (1, 2, 6, 8, 8, 8, 8, 4):        # uneven chunks, 45 tokens in total
        chunk = qkv[:, c0:c0 + q_len].reshape(B * q_len, -1).contiguous().cuda()
        ops.rope_append(chunk, None if cs is None else cs.cuda(), kc, vc, bt.cuda(), seq0, c0, B, q_len,
                        Hq, Hkv, D)
        q_out[:, c0:c0 + q_len] = chunk.float().cpu().view(B, q_len, -1)[:, :, :Hq * D]
        c0 += q_len
    assert c0 == n
    kcn, vcn = kc.float().cpu().numpy(), vc.float().cpu().numpy()
    half = D // 2
    for b in range(B):
        k_got, v_got = unpack_pages(kcn, vcn, bt[b].numpy(), n, Hkv, D)
        x = qkv[b].float().view(n, Hq + 2 * Hkv, D)
        q_ref, k_ref, v_ref = x[:, :Hq], x[:, Hq:Hq + Hkv], x[:, Hq + Hkv:]
        if rope:
            c, s = cs[:n, None, :half], cs[:n, None, half:]
            rot = lambda u: torch.cat([u[..., :half] * c - u[..., half:] * s,
                                       u[..., half:] * c + u[..., :half] * s], dim=-1)
            q_ref, k_ref = rot(q_ref), rot(k_ref)
        np.testing.assert_array_equal(v_got, v_ref.numpy())
        torch.testing.assert_close(torch.from_numpy(k_got), rb(k_ref), rtol=1e-2, atol=1e-2)
        torch.testing.assert_close(q_out[b].view(n, Hq, D), rb(q_ref), rtol=1e-2, atol=1e-2)


def test_attention_rejects_unsupported_shapes():
    from vllm_omni_draft_amd._lib import SdError
    ops = _ops()
    z = torch.zeros(8, 64 * 80, dtype=torch.bfloat16, device="cuda")
    bt = torch.zeros(1, 4, dtype=torch.int32, device="cuda")
    sl = torch.zeros(1, dtype=torch.int32, device="cuda")
    kc = torch.zeros(4, 1, 80 * 16, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(SdError):     # head_dim 80
        ops.paged_attention(z, kc, kc, bt, sl, 0, 1, 1, 4, 1, 80, 1.0, 8, 128)
    with pytest.raises(SdError):     # q_len * group > 64
        ops.paged_attention(z, kc, kc, bt, sl, 0, 1, 8, 64, 4, 64, 1.0, 8, 128)
    with pytest.raises(SdError):     # max_ctx beyond the block table
        ops.paged_attention(z, kc, kc, bt, sl, 0, 1, 1, 4, 1, 64, 1.0, 100, 128)


@pytest.mark.parametrize("B,ctx_lens,q_len,Hq,Hkv,D,split,N", [
    (1, [1100], 1, 32, 8, 64, 256, 2048),      # draft decode: 5 splits, half-tile projection
    (2, [700, 129], 2, 32, 8, 64, 256, 2048),  # draft first pass, ragged
    (1, [300], 1, 32, 8, 128, 128, 4096),      # target decode
    (1, [40], 2, 12, 12, 64, 128, 768),        # two splits bound, one live (MHA)
])
def test_projection_from_partials_equals_merge_then_linear(B, ctx_lens, q_len, Hq, Hkv, D, split, N):
    """Bit-exact: output projection fed by the attention partials == merge launch + linear."""
    ops = _ops()
    kc, vc, bt, ks, vs, qs = _setup(B, ctx_lens, q_len, Hq, Hkv, D, seed=11 + q_len)
    max_ctx = max(max(ctx_lens) + q_len, split + 1)
    pad = max(0, (max_ctx + 15) // 16 - bt.shape[1])
    bt = torch.cat([bt, torch.zeros(B, pad, dtype=torch.int32)], dim=1).contiguous()
    kc, vc, btc = kc.cuda(), vc.cuda(), bt.cuda()
    q = torch.cat(qs).reshape(B * q_len, Hq * D).cuda()
    seq_lens = torch.tensor(ctx_lens, dtype=torch.int32).cuda()
    gen = torch.Generator(device="cuda").manual_seed(N)
    W = ops.tile_weights((torch.randn(N, Hq * D, generator=gen, device="cuda") * 0.05).to(torch.bfloat16))
    res = torch.randn(B * q_len, N, generator=gen, device="cuda").to(torch.bfloat16)
    ws = torch.zeros(ops.paged_attention_workspace_bytes(B, q_len, Hq, D, max_ctx, split),
                     dtype=torch.uint8, device="cuda")
    attn = ops.paged_attention(q, kc, vc, btc, seq_lens, 0, B, q_len, Hq, Hkv, D, D ** -0.5, max_ctx,
                               split, workspace=ws)
    ref = ops.linear(W, attn, None, res, ops.EPI_RESIDUAL)
    ws.fill_(0xFF)    # stale partials of inactive splits must not matter (NaN bit patterns)
    assert ops.paged_attention(q, kc, vc, btc, seq_lens, 0, B, q_len, Hq, Hkv, D, D ** -0.5, max_ctx,
                               split, workspace=ws, partials_only=True) is None
    got = ops.linear_attn_merge(W, ws, seq_lens, 0, B, q_len, Hq, Hkv, D, max_ctx, split, None, res,
                                ops.EPI_RESIDUAL)
    assert torch.equal(got, ref)


@pytest.mark.parametrize("Hq,Hkv,D,q_total,q_block", [
    (32, 8, 128, 40, 16), (28, 4, 128, 31, 9), (8, 1, 128, 17, 8), (14, 2, 64, 64, 9), (4, 2, 64, 5, 5)])
@pytest.mark.parametrize("split,ctx_lens", [(128, [70, 0]), (256, [333, 18])])
def test_blocked_chunk_attention_equals_block_by_block_calls(Hq, Hkv, D, q_total, q_block, split,
                                                              ctx_lens):
    """A prompt chunk attended in one launch (blocks of q_block tokens as workgroup columns,
    ragged last block) is bit for bit what one call per block gives, and matches the oracle."""
    ops = _ops()
    B = len(ctx_lens)
    kc, vc, bt, ks, vs, qs = _setup(B, ctx_lens, q_total, Hq, Hkv, D, seed=7)
    kc, vc, btc = kc.cuda(), vc.cuda(), bt.cuda()
    q = torch.cat(qs).reshape(B * q_total, Hq * D).cuda()
    seq_lens = torch.tensor(ctx_lens, dtype=torch.int32).cuda()
    max_ctx = max(ctx_lens) + q_total
    got = ops.paged_attention_blocked(q, kc, vc, btc, seq_lens, 0, B, q_total, q_block, Hq, Hkv, D,
                                      D ** -0.5, max_ctx, split).view(B, q_total, Hq * D)
    for b in range(B):
        for t0 in range(0, q_total, q_block):
            n = min(q_block, q_total - t0)
            one = ops.paged_attention(q.view(B, q_total, -1)[b, t0:t0 + n].contiguous(), kc, vc,
                                      btc[b:b + 1].contiguous(), seq_lens[b:b + 1].contiguous(), t0, 1, n,
                                      Hq, Hkv, D, D ** -0.5, max_ctx, split)
            assert torch.equal(one, got[b, t0:t0 + n]), (b, t0)
        pos = [ctx_lens[b] + t for t in range(q_total)]
        ref = verify_attention(qs[b], ks[b], vs[b], pos, D ** -0.5)
        torch.testing.assert_close(got[b].float().cpu().view(q_total, Hq, D), ref, rtol=2e-2, atol=2e-2)


def test_blocked_attention_rejects_bad_blocks():
    ops = _ops()
    from vllm_omni_draft_amd import _lib
    kc, vc, bt, ks, vs, qs = _setup(1, [10], 8, 4, 2, 64, seed=1)
    q = torch.cat(qs).reshape(8, 4 * 64).cuda()
    sl = torch.tensor([10], dtype=torch.int32).cuda()
    with pytest.raises(_lib.SdError):   # block longer than the chunk
        ops.paged_attention_blocked(q, kc.cuda(), vc.cuda(), bt.cuda(), sl, 0, 1, 8, 9, 4, 2, 64, 0.125, 18, 128)
    with pytest.raises(_lib.SdError):   # 40 tokens * 2 heads per KV head > 64 rows
        ops.paged_attention_blocked(q, kc.cuda(), vc.cuda(), bt.cuda(), sl, 0, 1, 8, 40, 4, 2, 64, 0.125, 18, 128)
