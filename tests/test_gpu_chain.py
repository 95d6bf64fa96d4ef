"""The fused two-layer launches (sd_chain_o_up, sd_chain_down_qkv) against the launches they
replace.  The bar is bit-exactness: a fused launch keeps the summation order, roundings and
epilogues of linear_kernel, so every output row must equal the two-launch result bit for bit
(and the two-launch kernels are the ones checked against the oracle elsewhere)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ops():
    from vllm_omni_draft_amd import ops
    return ops


def _w(gen, rows, K, scale=None):
    return (torch.randn(rows, K, generator=gen, device="cuda") * (scale or K ** -0.5)).to(torch.bfloat16)


def _fp8(ops, W, mode, head_dim=0):
    q, s = ops.quantize_fp8(W)
    return ops.tile_weights(q, mode, head_dim, scale=s)


def _tile(ops, W, mode, head_dim=0, fp8=False):
    return _fp8(ops, W, mode, head_dim) if fp8 else ops.tile_weights(W, mode, head_dim)


# (Hq, Hkv, D, hidden, ffn): Llama-3.2-1B widths with a narrower MLP (fast), its full widths,
# and Llama-3-8B's (a draft for a 70B target)
SHAPES = [(32, 8, 64, 2048, 2048), (32, 8, 64, 2048, 8192), (32, 8, 128, 4096, 14336)]


@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("M", [1, 2, 4])
@pytest.mark.parametrize("Hq,Hkv,D,H,F", SHAPES)
def test_o_up_equals_output_projection_then_gated_mlp_input(Hq, Hkv, D, H, F, M, fp8):
    """Plain attention rows in, (h, act) out; three launches in a row on one state (the tags
    advance, h is fed back)."""
    ops = _ops()
    if not ops.chain_supported(M, H, Hq * D, F):
        pytest.skip("shape outside the fused form")
    gen = torch.Generator(device="cuda").manual_seed(H + F + M)
    o_w = _tile(ops, _w(gen, H, Hq * D), ops.TILE_PLAIN, fp8=fp8)
    up_w = _tile(ops, _w(gen, 2 * F, H), ops.TILE_GATED, fp8=fp8)
    g = (1 + 0.1 * torch.randn(H, generator=gen, device="cuda")).to(torch.bfloat16)
    h_ref = torch.randn(M, H, generator=gen, device="cuda").to(torch.bfloat16)
    h = h_ref.clone()
    state = ops.chain_state(H)
    for it in range(3):
        attn = torch.randn(M, Hq * D, generator=gen, device="cuda").to(torch.bfloat16)
        h_ref = ops.linear(o_w, attn, None, h_ref, ops.EPI_RESIDUAL)
        act_ref = ops.norm_linear(up_w, h_ref, g, None, 1e-5, False, None, ops.EPI_SILU_MUL)
        act = ops.chain_o_up(state, o_w, h, g, 1e-5, up_w, M, 1, Hq, Hkv, D, attn=attn)
        assert torch.equal(h, h_ref), f"launch {it}: h differs"
        assert torch.equal(act, act_ref), f"launch {it}: act differs"
    assert ops.chain_errors(state) == 0


@pytest.mark.parametrize("B,ctx_lens,q_len,split", [(1, [1100], 1, 256), (1, [1100], 2, 256),
                                                     (2, [700, 129], 2, 256), (1, [40], 1, 128)])
def test_o_up_merges_attention_partials_like_the_projection_launch(B, ctx_lens, q_len, split):
    ops = _ops()
    from tests.test_gpu_attention import _setup
    Hq, Hkv, D, H, F = 32, 8, 64, 2048, 2048
    kc, vc, bt, ks, vs, qs = _setup(B, ctx_lens, q_len, Hq, Hkv, D, seed=5 + q_len)
    max_ctx = max(max(ctx_lens) + q_len, split + 1)
    pad = max(0, (max_ctx + 15) // 16 - bt.shape[1])
    bt = torch.cat([bt, torch.zeros(B, pad, dtype=torch.int32)], dim=1).contiguous()
    kc, vc, btc = kc.cuda(), vc.cuda(), bt.cuda()
    q = torch.cat(qs).reshape(B * q_len, Hq * D).cuda()
    seq_lens = torch.tensor(ctx_lens, dtype=torch.int32).cuda()
    gen = torch.Generator(device="cuda").manual_seed(99)
    o_w = ops.tile_weights(_w(gen, H, Hq * D, 0.05))
    up_w = ops.tile_weights(_w(gen, 2 * F, H), ops.TILE_GATED)
    g = (1 + 0.1 * torch.randn(H, generator=gen, device="cuda")).to(torch.bfloat16)
    h0 = torch.randn(B * q_len, H, generator=gen, device="cuda").to(torch.bfloat16)
    ws = torch.zeros(ops.paged_attention_workspace_bytes(B, q_len, Hq, D, max_ctx, split),
                     dtype=torch.uint8, device="cuda")
    ws.fill_(0xFF)
    ops.paged_attention(q, kc, vc, btc, seq_lens, 0, B, q_len, Hq, Hkv, D, D ** -0.5, max_ctx, split,
                        workspace=ws, partials_only=True)
    h_ref = ops.linear_attn_merge(o_w, ws, seq_lens, 0, B, q_len, Hq, Hkv, D, max_ctx, split, None, h0,
                                  ops.EPI_RESIDUAL)
    act_ref = ops.norm_linear(up_w, h_ref, g, None, 1e-5, False, None, ops.EPI_SILU_MUL)
    state = ops.chain_state(H)
    h = h0.clone()
    act = ops.chain_o_up(state, o_w, h, g, 1e-5, up_w, B, q_len, Hq, Hkv, D, workspace=ws,
                         seq_lens=seq_lens, pos_offset=0, max_ctx=max_ctx, split_tokens=split)
    assert torch.equal(h, h_ref)
    assert torch.equal(act, act_ref)
    assert ops.chain_errors(state) == 0


@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("B,q_len", [(1, 1), (1, 2), (2, 2)])
@pytest.mark.parametrize("Hq,Hkv,D,H,F", SHAPES)
def test_down_qkv_equals_down_projection_then_next_layers_qkv(Hq, Hkv, D, H, F, B, q_len, fp8):
    ops = _ops()
    M = B * q_len
    if not ops.chain_supported(M, H, Hq * D, F):
        pytest.skip("shape outside the fused form")
    from vllm_omni_draft_amd.config import create_model_config, rope_table
    gen = torch.Generator(device="cuda").manual_seed(H + F + M)
    cfg = create_model_config(None, head_dim=D, n_heads=Hq, n_kv_heads=Hkv, max_pos=128)
    cs = rope_table(cfg).cuda()
    down_w = _tile(ops, _w(gen, H, F), ops.TILE_PLAIN, fp8=fp8)
    qkv_w = _tile(ops, _w(gen, (Hq + 2 * Hkv) * D, H), ops.TILE_ROTARY, D, fp8=fp8)
    g = (1 + 0.1 * torch.randn(H, generator=gen, device="cuda")).to(torch.bfloat16)
    bt = torch.tensor([[3, 1, 4], [0, 5, 2]], dtype=torch.int32, device="cuda")[:B].contiguous()
    seq_lens = torch.tensor([17, 30][:B], dtype=torch.int32, device="cuda")
    h_ref = torch.randn(M, H, generator=gen, device="cuda").to(torch.bfloat16)
    h = h_ref.clone()
    state = ops.chain_state(H)
    for it in range(2):
        act = torch.randn(M, F, generator=gen, device="cuda").to(torch.bfloat16)
        kc_ref = torch.zeros(6, Hkv, D * 16, dtype=torch.bfloat16, device="cuda")
        vc_ref, kc, vc = torch.zeros_like(kc_ref), torch.zeros_like(kc_ref), torch.zeros_like(kc_ref)
        h_ref = ops.linear(down_w, act, None, h_ref, ops.EPI_RESIDUAL)
        q_ref = ops.norm_qkv_rope(qkv_w, h_ref, g, None, 1e-5, False, None, cs, kc_ref, vc_ref, bt,
                                  seq_lens, -1, B, q_len, Hq, Hkv, D)
        q = ops.chain_down_qkv(state, down_w, act, h, g, 1e-5, qkv_w, cs, kc, vc, bt, seq_lens, -1, B,
                               q_len, Hq, Hkv, D)
        assert torch.equal(h, h_ref), f"launch {it}: h differs"
        assert torch.equal(q, q_ref), f"launch {it}: q differs"
        assert torch.equal(kc, kc_ref) and torch.equal(vc, vc_ref), f"launch {it}: cache differs"
        assert kc.abs().sum() > 0 and vc.abs().sum() > 0
    assert ops.chain_errors(state) == 0


def test_unsupported_shapes_are_refused():
    ops = _ops()
    from vllm_omni_draft_amd._lib import SdError
    assert not ops.chain_supported(5, 2048, 2048, 8192)       # rows
    assert not ops.chain_supported(1, 896, 896, 4864)         # widths
    assert ops.chain_supported(2, 2048, 2048, 8192)
    gen = torch.Generator(device="cuda").manual_seed(1)
    o_w = ops.tile_weights(_w(gen, 896, 896))
    up_w = ops.tile_weights(_w(gen, 2 * 1024, 896), ops.TILE_GATED)
    h = torch.zeros(1, 896, dtype=torch.bfloat16, device="cuda")
    attn = torch.zeros(1, 896, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(SdError):
        ops.chain_o_up(ops.chain_state(896), o_w, h, h[0], 1e-5, up_w, 1, 1, 14, 2, 64, attn=attn)


def _draft_sized(n_layers=3, fp8=False):
    """A rotary model with Llama-3.2-1B's layer widths (hidden 2048, 32/8 heads of 64) and a
    narrow MLP / vocabulary so that it builds in a second."""
    from oracle.models import OracleConfig, random_state
    from tests.test_gpu_model import _bf, _product
    oc = OracleConfig(arch="llama", n_layers=n_layers, hidden=2048, n_heads=32, n_kv_heads=8, head_dim=64,
                      ffn=2048, vocab=512, max_pos=2048)
    return _product(oc, _bf(random_state(oc, 77)), fp8=fp8)


@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("ctx,split", [(40, 128), (300, 128), (700, 256)])
def test_model_passes_with_fused_pairs_have_the_same_logits_and_cache(ctx, split, fp8):
    """sd_model_forward with a chain state: one split (plain attention rows), several splits
    (partials merged by the fused launch), passes of 2 rows and of 1 row, against the same
    model without the state; logits and the cache it leaves behind bit for bit."""
    from vllm_omni_draft_amd.kv_cache import PagedKVCache
    from vllm_omni_draft_amd.model_runner import DecoderModel
    w = _draft_sized(fp8=fp8)
    cfg = w.cfg
    gen = torch.Generator().manual_seed(ctx)
    tokens = torch.randint(0, cfg.vocab, (ctx + 6,), generator=gen, dtype=torch.int32).cuda()
    results = []
    for fused in (False, True):
        model = DecoderModel(w, max_ctx=1024, split_tokens=split, fused_pairs=fused)
        kv = PagedKVCache(cfg, 80, 1, 1024, "cuda")
        kv.ensure_capacity(0, ctx + 6)
        zero = torch.zeros(1, dtype=torch.int32, device="cuda")
        c0 = 0
        while c0 < ctx:                              # the prompt, in wide passes
            q = min(256, ctx - c0)
            model.forward(kv, tokens, 0, c0, zero, c0, 1, q, None, 1024, wide=True)
            c0 += q
        outs = []
        for q in (2, 1, 1, 2):                       # a draft's decode passes
            lg = torch.empty(q, cfg.vocab, dtype=torch.float32, device="cuda")
            model.forward(kv, tokens, 0, c0, zero, c0, 1, q, lg, c0 + q)
            outs.append(lg)
            c0 += q
        torch.cuda.synchronize()
        if fused:
            from vllm_omni_draft_amd import ops
            assert ops.chain_errors(model.chain_state) == 0
        results.append((torch.cat(outs), kv.k.clone(), kv.v.clone()))
    (la, ka, va), (lb, kb, vb) = results
    assert torch.isfinite(la).all() and la.abs().max() > 0
    assert torch.equal(la, lb)
    assert torch.equal(ka, kb) and torch.equal(va, vb)


def test_engine_with_fused_draft_passes_emits_the_same_stream_and_accepts_the_same():
    from vllm_omni_draft_amd.engine import EngineConfig, SpecEngine
    tw, dw = _draft_sized(3), _draft_sized(2)
    prompt = [int(t) for t in torch.randint(0, 512, (150,), generator=torch.Generator().manual_seed(3))]
    runs = []
    for fused in (False, True):
        for graph in (False, True):
            eng = SpecEngine(tw, dw, EngineConfig(k=3, max_seq_len=512, split_tokens=128,
                                                  use_graph=graph, draft_fused_pairs=fused))
            out = [eng.add_request(0, prompt)]
            while len(out) < 40:
                out.extend(eng.step()[0])
            m = eng.metrics()
            runs.append((out[:40], m["accepted"], m["draft"]))
            if fused:
                from vllm_omni_draft_amd import ops
                assert ops.chain_errors(eng.draft.chain_state) == 0
    assert all(r == runs[0] for r in runs[1:]), runs
