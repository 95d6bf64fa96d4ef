"""The draft model's three-launch layer (sd_norm_qkv_rope_adds, sd_draft_attn_oproj,
sd_draft_mlp): every launch against the CPU oracle or against the library's own oracle-tested
kernels composed the long way, whole draft passes against the oracle's logits, and engines whose
draft runs on it: the emitted stream stays the target's greedy stream, whatever the draft's
rounding."""
import pytest
import torch

from oracle.attention import verify_attention
from oracle.models import OracleConfig, OracleLM, random_state
from tests.test_gpu_attention import _setup

pytestmark = pytest.mark.gpu


def _ops():
    from vllm_omni_draft_amd import ops
    return ops


ATTN_CASES = [
    # B, ctx_lens, q_len, Hq, Hkv, D, H
    (1, [0], 1, 32, 8, 64, 2048),
    (1, [15], 2, 32, 8, 64, 2048),
    (1, [1100], 1, 32, 8, 64, 2048),          # the headline draft's decode pass
    (1, [1100], 2, 32, 8, 64, 2048),          # and its first pass of a step (two tokens)
    (2, [2040, 37], 2, 32, 8, 64, 2048),      # up to the context limit, ragged batch
    (3, [300, 31, 257], 1, 14, 2, 64, 896),   # 7 query heads per KV head, 14 projection k-steps
    (1, [700], 2, 8, 2, 128, 1024),           # head_dim 128
    (4, [5, 64, 129, 512], 1, 4, 4, 64, 256), # one query head per KV head
]


@pytest.mark.parametrize("B,ctx_lens,q_len,Hq,Hkv,D,H", ATTN_CASES)
def test_attention_with_output_projection_matches_the_oracle(B, ctx_lens, q_len, Hq, Hkv, D, H):
    """Launch 2: the Hkv partial rows, summed, are o_proj(attention); the attention part is the
    CPU oracle's, the projection an f32 matmul on its bf16-rounded rows.  Every slab is the
    product with its own head group's columns.  The accumulator handed in is cleared."""
    ops = _ops()
    kc, vc, bt, ks, vs, qs = _setup(B, ctx_lens, q_len, Hq, Hkv, D, seed=sum(ctx_lens) + q_len + H)
    g = torch.Generator().manual_seed(H + q_len)
    Wo = (torch.randn(H, Hq * D, generator=g) * 0.05).to(torch.bfloat16)
    q = torch.cat(qs).reshape(B * q_len, Hq * D).cuda()
    seq_lens = torch.tensor(ctx_lens, dtype=torch.int32).cuda()
    max_ctx = max(ctx_lens) + q_len
    dirty = torch.full((B * q_len * H,), 3.0, dtype=torch.float32, device="cuda")
    slabs = ops.draft_attn_oproj(q, kc.cuda(), vc.cuda(), bt.cuda(), seq_lens, 0, ops.tile_weights(Wo.cuda()),
                                 B, q_len, Hq, Hkv, D, D ** -0.5, max_ctx, zero=dirty)
    torch.cuda.synchronize()
    assert slabs.shape == (Hkv, B * q_len, H) and not dirty.any()
    slabs = slabs.cpu()
    G = Hq // Hkv
    for b in range(B):
        pos = [ctx_lens[b] + t for t in range(q_len)]
        attn = verify_attention(qs[b], ks[b], vs[b], pos, D ** -0.5)            # [q_len, Hq, D] f32
        attn = attn.to(torch.bfloat16).float().reshape(q_len, Hq * D)
        want = attn @ Wo.float().T
        got = slabs[:, b * q_len:(b + 1) * q_len].sum(0)
        scale = want.abs().max().item()
        assert (got - want).abs().max().item() < 0.02 * scale
        for kvh in range(Hkv):
            cols = slice(kvh * G * D, (kvh + 1) * G * D)
            part = attn[:, cols] @ Wo.float()[:, cols].T
            assert (slabs[kvh, b * q_len:(b + 1) * q_len] - part).abs().max().item() < 0.02 * scale


def test_attention_with_output_projection_refuses_what_it_cannot_do():
    from vllm_omni_draft_amd import _lib
    ops = _ops()
    assert ops.draft_layer_supported(1, 1, 2048, 32, 8, 64, 8192, 2048)
    assert ops.draft_layer_supported(4, 2, 896, 14, 2, 64, 4864, 1200)
    assert not ops.draft_layer_supported(1, 1, 2048, 32, 8, 64, 8192, 2049)     # context beyond one workgroup
    assert not ops.draft_layer_supported(5, 1, 2048, 32, 8, 64, 8192, 512)      # rows
    assert not ops.draft_layer_supported(3, 3, 2048, 32, 8, 64, 8192, 512)      # tokens per sequence
    assert not ops.draft_layer_supported(1, 1, 2000, 32, 8, 64, 8192, 512)      # hidden % 128
    assert not ops.draft_layer_supported(1, 1, 2048, 32, 8, 80, 8192, 512)      # head_dim
    kc, vc, bt, ks, vs, qs = _setup(1, [40], 1, 32, 8, 64, seed=1)
    Wo = ops.tile_weights(torch.zeros(2048, 2048, dtype=torch.bfloat16, device="cuda"))
    q = torch.cat(qs).reshape(1, 2048).cuda()
    sl = torch.tensor([40], dtype=torch.int32).cuda()
    with pytest.raises(_lib.SdError):
        ops.draft_attn_oproj(q, kc.cuda(), vc.cuda(), bt.cuda(), sl, 0, Wo, 1, 1, 32, 8, 64, 0.125, 4096)


@pytest.mark.parametrize("M,H,F,n_add", [(1, 2048, 8192, 8), (2, 2048, 8192, 8), (4, 2048, 8192, 8),
                                         (1, 896, 4864, 2), (3, 1024, 2048, 0), (2, 4096, 1024, 1),
                                         (1, 128, 64, 4)])
def test_mlp_launch_matches_the_two_kernels_it_replaces(M, H, F, n_add):
    """Launch 3 against the oracle-tested kernels composed the long way on the same input:
    x = bf16(h + slabs), sd_norm_linear (gate/up, SiLU * mul), sd_linear (down).  The
    activations are the same bits; the down projection is summed in another order (f32
    atomics over ffn/32 partial rows), hence a tolerance.  Repeated launches keep adding."""
    ops = _ops()
    g = torch.Generator().manual_seed(M * 7 + H + n_add)
    h = torch.randn(M, H, generator=g).to(torch.bfloat16).cuda()
    add = (torch.randn(n_add, M, H, generator=g) * 0.3).cuda() if n_add else None
    norm_w = (1.0 + 0.1 * torch.randn(H, generator=g)).to(torch.bfloat16).cuda()
    Wup = (torch.randn(2 * F, H, generator=g) * 0.03).to(torch.bfloat16).cuda()
    Wdown = (torch.randn(H, F, generator=g) * 0.03).to(torch.bfloat16).cuda()
    up_t, down_t = ops.tile_weights(Wup, ops.TILE_GATED), ops.tile_weights(Wdown)
    x = h.float()
    if n_add:
        s = add[0].clone()
        for i in range(1, n_add):
            s += add[i]
        x = x + s
    x = x.to(torch.bfloat16)
    act = ops.norm_linear(up_t, x, norm_w, None, 1e-5, False, None, ops.EPI_SILU_MUL)
    want = ops.linear(down_t, act, None, None, ops.EPI_F32).float()
    acc0 = torch.randn(M, H, generator=g).cuda()
    acc = acc0.clone()
    h_out = torch.zeros_like(h)
    ops.draft_mlp(h, add, norm_w, 1e-5, up_t, ops.kmajor_tiles(down_t), acc, F, h_out=h_out)
    torch.cuda.synchronize()
    assert torch.equal(h_out, x)
    scale = want.abs().max().item()
    assert ((acc - acc0) - want).abs().max().item() < 5e-3 * scale + 1e-5
    ops.draft_mlp(h, add, norm_w, 1e-5, up_t, ops.kmajor_tiles(down_t), acc, F)
    torch.cuda.synchronize()
    assert ((acc - acc0) - 2 * want).abs().max().item() < 1e-2 * scale + 1e-5


@pytest.mark.parametrize("M,q_len", [(1, 1), (2, 2), (4, 2), (3, 1)])
def test_fused_norm_launches_with_an_additive_input(M, q_len):
    """sd_norm_qkv_rope_adds / sd_norm_linear_adds on (x, add) give exactly what the plain
    launches give on bf16(x + add): q, the cache entries, the logits; x_out holds the sum."""
    from vllm_omni_draft_amd import config
    ops = _ops()
    Hq, Hkv, D, K, V = 8, 2, 64, 1024, 4096
    B = M // q_len
    g = torch.Generator().manual_seed(M)
    W = (torch.randn((Hq + 2 * Hkv) * D, K, generator=g) * 0.05).to(torch.bfloat16).cuda()
    Wt = ops.tile_weights(W, ops.TILE_ROTARY, D)
    head = ops.tile_weights((torch.randn(V, K, generator=g) * 0.05).to(torch.bfloat16).cuda())
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    add = (torch.randn(M, K, generator=g) * 0.5).cuda()
    norm_w = (1.0 + 0.1 * torch.randn(K, generator=g)).to(torch.bfloat16).cuda()
    cs = config.rope_table(config.create_model_config(None, head_dim=D, max_pos=128), "cuda")
    bt = torch.arange(B * 4, dtype=torch.int32, device="cuda").view(B, 4)
    sl = torch.tensor([5 + 9 * b for b in range(B)], dtype=torch.int32, device="cuda")
    xs = (x.float() + add).to(torch.bfloat16)

    def caches():
        return (torch.zeros(B * 4, Hkv, 16 * D, dtype=torch.bfloat16, device="cuda"),
                torch.zeros(B * 4, Hkv, 16 * D, dtype=torch.bfloat16, device="cuda"))
    kc0, vc0 = caches()
    q0 = ops.norm_qkv_rope(Wt, xs, norm_w, None, 1e-5, False, None, cs, kc0, vc0, bt, sl, 0, B, q_len, Hq, Hkv, D)
    kc1, vc1 = caches()
    x_out = torch.zeros_like(x)
    q1 = ops.norm_qkv_rope_adds(Wt, x, norm_w, 1e-5, cs, kc1, vc1, bt, sl, 0, B, q_len, Hq, Hkv, D,
                                add=add, x_out=x_out)
    torch.cuda.synchronize()
    assert torch.equal(q0, q1) and torch.equal(kc0, kc1) and torch.equal(vc0, vc1) and torch.equal(x_out, xs)
    kc2, vc2 = caches()
    q2 = ops.norm_qkv_rope_adds(Wt, xs, norm_w, 1e-5, cs, kc2, vc2, bt, sl, 0, B, q_len, Hq, Hkv, D)   # no add
    assert torch.equal(q0, q2) and torch.equal(kc0, kc2)
    want = ops.norm_linear(head, xs, norm_w, None, 1e-5, False, None, ops.EPI_F32)
    got = ops.norm_linear_adds(head, x, norm_w, 1e-5, add=add)
    torch.cuda.synchronize()
    assert torch.equal(want, got)


DRAFTS = {
    "llama-3.2-1b-width": OracleConfig(arch="llama", n_layers=2, hidden=2048, n_heads=32, n_kv_heads=8,
                                       head_dim=64, ffn=8192, vocab=4096, rope_theta=500000.0, max_pos=512,
                                       tie_embeddings=True),
    "small": OracleConfig(arch="llama", n_layers=3, hidden=256, n_heads=4, n_kv_heads=2, head_dim=64,
                          ffn=512, vocab=512, max_pos=512),
    "qwen-0.5b-width-no-bias": OracleConfig(arch="llama", n_layers=2, hidden=896, n_heads=14, n_kv_heads=2,
                                            head_dim=64, ffn=4864, vocab=1024, norm_eps=1e-6, rope_theta=1e6,
                                            max_pos=512, tie_embeddings=True),
}


@pytest.mark.parametrize("name", list(DRAFTS))
def test_whole_passes_on_the_three_launch_layer_match_the_oracle(name):
    """A model whose passes run the three-launch layer: logits of single-token and two-token
    passes behind a prefilled prompt against the oracle (the tolerance of the five-launch
    forward) and against the five-launch forward itself (closer still, same arg-max)."""
    from tests.test_gpu_model import _bf, _product
    from vllm_omni_draft_amd.kv_cache import PagedKVCache
    from vllm_omni_draft_amd.model_runner import DecoderModel
    oc = DRAFTS[name]
    state = _bf(random_state(oc, seed=61))
    w = _product(oc, state)
    toks = torch.randint(0, oc.vocab, (31,), generator=torch.Generator().manual_seed(5)).tolist()
    tok = torch.tensor(toks, dtype=torch.int32, device="cuda")
    zero = torch.zeros(1, dtype=torch.int32, device="cuda")
    chunks = [8, 8, 4, 1, 1, 2, 1, 2, 2, 1, 1]          # the prompt in wide-ish pieces, then 1 / 2 rows

    def run(fused):
        model = DecoderModel(w, max_ctx=128, split_tokens=128, fused_layers=fused)
        assert model.fused_layers == fused
        kv = PagedKVCache(w.cfg, 16, 1, 128, "cuda")
        kv.ensure_capacity(0, len(toks))
        outs, c0 = [], 0
        for q in chunks:
            lg = torch.empty(q, oc.vocab, dtype=torch.float32, device="cuda")
            model.forward(kv, tok, 0, c0, zero, c0, 1, q, lg, 128)
            outs.append(lg)
            c0 += q
        torch.cuda.synchronize()
        return torch.cat(outs).cpu()

    got, plain = run(True), run(False)
    lm = OracleLM(oc, state, bf16_acts=True)
    ref = lm.forward(toks, 0, lm.new_cache())
    scale = ref.abs().max().item()
    assert (got - ref).abs().max().item() < 0.03 * scale, (got - ref).abs().max().item() / scale
    assert (got.argmax(-1) == ref.argmax(-1)).float().mean().item() > 0.9
    assert torch.equal(got[:20], plain[:20])            # passes of more than 4 rows: the same launches
    assert (got - plain).abs().max().item() < 0.02 * scale
    assert (got[20:].argmax(-1) == plain[20:].argmax(-1)).float().mean().item() >= 0.8


@pytest.mark.parametrize("use_graph", [False, True])
@pytest.mark.parametrize("B", [1, 2])
def test_engines_with_a_three_launch_draft_stay_lossless(B, use_graph):
    """Greedy speculative decoding with the draft on the three-launch layer emits the target's
    autoregressive stream token for token (k = 5, contexts crossing pages), proposals are
    accepted as often as with the five-launch draft up to a few near-ties, and a draft that
    IS the target has nearly every proposal accepted."""
    from tests.test_gpu_model import _bf, _generate, _product
    from vllm_omni_draft_amd.engine import EngineConfig, SpecEngine
    tcfg = OracleConfig(arch="llama", n_layers=2, hidden=256, n_heads=4, n_kv_heads=2, head_dim=64,
                        ffn=512, vocab=512, max_pos=512)
    tw = _product(tcfg, _bf(random_state(tcfg, seed=71)))
    dw = _product(DRAFTS["small"], _bf(random_state(DRAFTS["small"], seed=72)))
    prompts = [[5, 9, 200, 31, 7], list(range(40, 73))][:B]

    def engine(draft, k, **kw):
        return SpecEngine(tw, draft, EngineConfig(k=k, max_batch=B, max_seq_len=300, split_tokens=128,
                                                  use_graph=use_graph, rows="narrow", **kw))
    ar = _generate(engine(None, 0), prompts, 120)
    fused = engine(dw, 5, draft_fused_layers=True)
    assert fused.draft.fused_layers
    assert _generate(fused, prompts, 120) == ar
    plain = engine(dw, 5)
    assert not plain.draft.fused_layers
    assert _generate(plain, prompts, 120) == ar
    mf, mp = fused.metrics(), plain.metrics()
    assert mf["draft"] > 0 and abs(mf["accepted"] - mp["accepted"]) <= 0.1 * max(mp["accepted"], 10)
    same = engine(tw, 5, draft_fused_layers=True)   # the target as its own draft, through the other layer
    assert same.draft.fused_layers
    assert _generate(same, prompts, 120) == ar
    m = same.metrics()
    assert m["accepted"] >= 0.9 * m["draft"]
