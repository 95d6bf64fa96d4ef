"""Whole-model forward passes (native runtime) against the CPU oracle, plus the
engine-level invariant: greedy speculative decoding == greedy autoregressive
decoding through the same kernels, token for token."""
import json
import os

import numpy as np
import pytest
import torch

from oracle.models import OracleConfig, OracleLM, random_state
from oracle.spec_decode import ar_generate, spec_generate

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _product(ocfg: OracleConfig, state, qkv_bias=False, fp8=False):
    from vllm_omni_draft_amd import config, weights
    cfg = config.create_model_config(
        None, arch=ocfg.arch, n_layers=ocfg.n_layers, hidden=ocfg.hidden, n_heads=ocfg.n_heads,
        n_kv_heads=ocfg.n_kv_heads, head_dim=ocfg.head_dim, ffn=ocfg.ffn, vocab=ocfg.vocab,
        norm_eps=ocfg.norm_eps, rope_theta=ocfg.rope_theta, rope_scaling=ocfg.rope_scaling,
        max_pos=ocfg.max_pos, tie_embeddings=ocfg.tie_embeddings, qkv_bias=qkv_bias)
    return weights.from_state_dict(cfg, state, "cuda", fp8=fp8)


def _fp8_dequantised(ocfg: OracleConfig, state):
    """The state dict the fp8 product computes with: every projection matrix (and the LM
    head, a quantised copy when tied) replaced by fp8(W / scale) * scale, per-row scales."""
    from vllm_omni_draft_amd import ops
    out = {}
    for k, v in state.items():
        is_matrix = k.endswith("weight") and v.dim() == 2 and "embed" not in k
        if is_matrix:
            q, sc = ops.quantize_fp8(v.to(torch.bfloat16))
            v = q.float() * sc[:, None]
        out[k] = v
    if "lm_head.weight" not in state:
        emb = next(v for k, v in state.items() if k.endswith("embed_tokens.weight"))
        q, sc = ops.quantize_fp8(emb.to(torch.bfloat16))
        out["lm_head.weight"] = q.float() * sc[:, None]
    return out


def _bf(state):
    return {k: v.to(torch.bfloat16).float() for k, v in state.items()}


def _logits_via_engine(w, tokens, chunks, wide=False, read_ahead_mb=0, fp8_activations=False):
    """Feeds `tokens` through the native forward in the given chunk sizes."""
    from vllm_omni_draft_amd.kv_cache import PagedKVCache
    from vllm_omni_draft_amd.model_runner import DecoderModel
    cfg = w.cfg
    model = DecoderModel(w, max_ctx=128, split_tokens=128, wide=wide, read_ahead_mb=read_ahead_mb,
                         fp8_activations=fp8_activations)
    kv = PagedKVCache(cfg, 16, 1, 128, "cuda")
    kv.ensure_capacity(0, len(tokens))
    tok = torch.tensor(tokens, dtype=torch.int32, device="cuda")
    zero = torch.zeros(1, dtype=torch.int32, device="cuda")
    outs, c0 = [], 0
    for q in chunks:
        lg = torch.empty(q, cfg.vocab, dtype=torch.float32, device="cuda")
        model.forward(kv, tok, 0, c0, zero, c0, 1, q, lg, 128)
        outs.append(lg)
        c0 += q
    assert c0 == len(tokens)
    torch.cuda.synchronize()
    return torch.cat(outs).cpu()


MODELS = {
    "llama": OracleConfig(arch="llama", n_layers=2, hidden=128, n_heads=4, n_kv_heads=2, head_dim=64,
                          ffn=256, vocab=256, max_pos=256),
    "llama3_rope_gqa4": OracleConfig(arch="llama", n_layers=2, hidden=256, n_heads=4, n_kv_heads=1,
                                     head_dim=128, ffn=320, vocab=320, rope_theta=500000.0,
                                     rope_scaling={"rope_type": "llama3", "factor": 8.0,
                                                   "low_freq_factor": 1.0, "high_freq_factor": 4.0,
                                                   "original_max_position_embeddings": 32},
                                     max_pos=256, tie_embeddings=True),
    "qwen2": OracleConfig(arch="llama", n_layers=2, hidden=448, n_heads=7, n_kv_heads=1, head_dim=64,
                          ffn=608, vocab=512, norm_eps=1e-6, rope_theta=1e6, max_pos=256),
    "opt": OracleConfig(arch="opt", n_layers=2, hidden=192, n_heads=3, n_kv_heads=3, head_dim=64,
                        ffn=384, vocab=272, max_pos=256, tie_embeddings=True),
}


@pytest.mark.parametrize("name", list(MODELS))
def test_forward_logits_match_oracle(name):
    ocfg = MODELS[name]
    state = _bf(random_state(ocfg, seed=11, qkv_bias=(name == "qwen2")))
    w = _product(ocfg, state, qkv_bias=(name == "qwen2"))
    toks = torch.randint(0, ocfg.vocab, (37,), generator=torch.Generator().manual_seed(4)).tolist()
    got = _logits_via_engine(w, toks, [8, 8, 8, 6, 1, 2, 4])
    lm = OracleLM(ocfg, state, bf16_acts=True)
    ref = lm.forward(toks, 0, lm.new_cache())
    # bf16 activations: compare on the scale of the logits
    scale = ref.abs().max().item()
    assert (got - ref).abs().max().item() < 0.03 * scale, (got - ref).abs().max().item() / scale
    agree = (got.argmax(-1) == ref.argmax(-1)).float().mean().item()
    assert agree > 0.9


_LLAMA3_SCALING = {"rope_type": "llama3", "factor": 8.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
                   "original_max_position_embeddings": 8192}
FULL_WIDTH = {
    # BASELINE configs[1]: Llama-3-8B target, Llama-3.2-1B draft
    "llama3-8b+1b": (
        OracleConfig(arch="llama", n_layers=2, hidden=4096, n_heads=32, n_kv_heads=8, head_dim=128,
                     ffn=14336, vocab=128256, rope_theta=500000.0, rope_scaling=_LLAMA3_SCALING,
                     max_pos=512, tie_embeddings=True),
        OracleConfig(arch="llama", n_layers=2, hidden=2048, n_heads=32, n_kv_heads=8, head_dim=64,
                     ffn=8192, vocab=128256, rope_theta=500000.0, rope_scaling=_LLAMA3_SCALING,
                     max_pos=512, tie_embeddings=True), False),
    # BASELINE configs[2]: Qwen2.5-7B target (7 query heads per KV head, QKV biases), 0.5B draft
    "qwen2.5-7b+0.5b": (
        OracleConfig(arch="llama", n_layers=2, hidden=3584, n_heads=28, n_kv_heads=4, head_dim=128,
                     ffn=18944, vocab=152064, norm_eps=1e-6, rope_theta=1e6, max_pos=512,
                     tie_embeddings=True),
        OracleConfig(arch="llama", n_layers=2, hidden=896, n_heads=14, n_kv_heads=2, head_dim=64,
                     ffn=4864, vocab=152064, norm_eps=1e-6, rope_theta=1e6, max_pos=512,
                     tie_embeddings=True), True),
}


@pytest.mark.parametrize("wide", [False, True])
@pytest.mark.parametrize("pair", list(FULL_WIDTH))
def test_full_width_layers_against_the_oracle(pair, wide):
    """The named configurations' layer shapes at full width (hidden size, head grouping, ffn,
    vocabulary, rope variant, biases), two layers of each so that the CPU oracle can follow, on
    both kernel families: logits of prompt chunks, decode tokens and a 6-row verify pass against
    the oracle, and greedy speculative == autoregressive through the engine."""
    tcfg, dcfg, bias = FULL_WIDTH[pair]
    tstate = _bf(random_state(tcfg, seed=31, qkv_bias=bias))
    tw = _product(tcfg, tstate, qkv_bias=bias)
    toks = torch.randint(0, tcfg.vocab, (20,), generator=torch.Generator().manual_seed(9)).tolist()
    got = _logits_via_engine(tw, toks, [8, 1, 1, 6, 4], wide=wide)
    lm = OracleLM(tcfg, tstate, bf16_acts=True)
    ref = lm.forward(toks, 0, lm.new_cache())
    scale = ref.abs().max().item()
    assert (got - ref).abs().max().item() < 0.03 * scale, (got - ref).abs().max().item() / scale
    assert (got.argmax(-1) == ref.argmax(-1)).float().mean().item() >= 0.8
    del lm, ref, tstate
    dw = _product(dcfg, _bf(random_state(dcfg, seed=32, qkv_bias=bias)), qkv_bias=bias)
    rows = "wide" if wide else "narrow"
    ar = _generate(_engine(tw, None, 0, rows=rows), [toks], 24)[0]
    eng = _engine(tw, dw, 5, rows=rows)
    assert _generate(eng, [toks], 24)[0] == ar
    m = eng.metrics()
    assert m["draft"] > 0 and m["emitted"] >= 24 - 1


@pytest.mark.parametrize("which", [0, 1])
def test_weight_read_ahead_in_the_attention_launch_changes_no_bit(which):
    """The read-ahead workgroups behind the attention grid only load: logits of prompt chunks,
    decode tokens and a 6-row pass at Llama-3-8B / 3.2-1B width (8 KV heads: the workgroups know
    their XCD) equal those of the plain launches bit for bit, for caps that end inside the
    output projection, inside gate/up, and past both."""
    cfg = FULL_WIDTH["llama3-8b+1b"][which]
    w = _product(cfg, _bf(random_state(cfg, seed=33)))
    toks = torch.randint(0, cfg.vocab, (20,), generator=torch.Generator().manual_seed(10)).tolist()
    ref = _logits_via_engine(w, toks, [8, 1, 1, 6, 4])
    for mb in (1, 24, 4096):
        got = _logits_via_engine(w, toks, [8, 1, 1, 6, 4], read_ahead_mb=mb)
        assert torch.equal(got, ref), mb


@pytest.mark.parametrize("name", list(MODELS))
def test_wide_forward_logits_match_oracle_and_are_chunk_invariant(name):
    """The 17..256-row kernel family: same oracle bar, and a token's logits do not depend on
    the number of rows in its pass (within the family)."""
    ocfg = MODELS[name]
    state = _bf(random_state(ocfg, seed=11, qkv_bias=(name == "qwen2")))
    w = _product(ocfg, state, qkv_bias=(name == "qwen2"))
    toks = torch.randint(0, ocfg.vocab, (37,), generator=torch.Generator().manual_seed(4)).tolist()
    G = ocfg.n_heads // ocfg.n_kv_heads
    big = min(64 // G, 37)                      # attention bound: q_len * G <= 64
    chunks = [big] * (37 // big) + ([37 % big] if 37 % big else [])
    got = _logits_via_engine(w, toks, chunks, wide=True)
    lm = OracleLM(ocfg, state, bf16_acts=True)
    ref = lm.forward(toks, 0, lm.new_cache())
    scale = ref.abs().max().item()
    assert (got - ref).abs().max().item() < 0.03 * scale, (got - ref).abs().max().item() / scale
    assert (got.argmax(-1) == ref.argmax(-1)).float().mean().item() > 0.9
    again = _logits_via_engine(w, toks, [1] * 5 + [8, 8, 8, 6, 2], wide=True)
    assert torch.equal(got, again)


@pytest.mark.parametrize("name", ["llama", "opt", "qwen2"])
def test_fp8_forward_logits_match_oracle_on_dequantised_weights(name):
    """fp8 weights (quantised at load, per-row scales): the native forward against the CPU
    oracle run on the dequantised matrices, to the bf16-activation tolerance of the bf16 test."""
    ocfg = MODELS[name]
    if ocfg.ffn % 64:   # fp8 tiles hold 64 input columns
        ocfg = OracleConfig(**{**vars(ocfg), "ffn": (ocfg.ffn + 63) // 64 * 64})
    state = _bf(random_state(ocfg, seed=11, qkv_bias=(name == "qwen2")))
    w = _product(ocfg, state, qkv_bias=(name == "qwen2"), fp8=True)
    assert w.fp8 and w.lm_head.data.dtype == torch.uint8 and not w.embed_tiled
    toks = torch.randint(0, ocfg.vocab, (37,), generator=torch.Generator().manual_seed(4)).tolist()
    got = _logits_via_engine(w, toks, [8, 8, 8, 6, 1, 2, 4])
    lm = OracleLM(ocfg, _fp8_dequantised(ocfg, state), bf16_acts=True)
    ref = lm.forward(toks, 0, lm.new_cache())
    scale = ref.abs().max().item()
    assert (got - ref).abs().max().item() < 0.03 * scale, (got - ref).abs().max().item() / scale
    assert (got.argmax(-1) == ref.argmax(-1)).float().mean().item() > 0.9
    # and it is a different model from the bf16 one (the quantisation is really applied)
    plain = _logits_via_engine(_product(ocfg, state, qkv_bias=(name == "qwen2")), toks, [8, 8, 8, 6, 1, 2, 4])
    assert (got - plain).abs().max().item() > 1e-3 * scale


@pytest.mark.parametrize("use_graph", [False, True])
def test_fp8_greedy_spec_equals_ar(use_graph):
    """The engine invariant holds with fp8 weights on both models, alone (narrow family) and
    in a batch of 3 (wide family for the target, 192-token prefill passes in both)."""
    tcfg, dcfg = MODELS["llama"], MODELS["opt"]
    dcfg = OracleConfig(**{**vars(dcfg), "vocab": tcfg.vocab})
    tw = _product(tcfg, _bf(random_state(tcfg, seed=21)), fp8=True)
    dw = _product(dcfg, _bf(random_state(dcfg, seed=22)), fp8=True)
    prompts = [[5, 9, 200, 31, 7], list(range(40, 73)), [1, 2, 3]]
    ar = [_generate(_engine(tw, None, 0, use_graph=use_graph), [p], 40)[0] for p in prompts]
    for p, want in zip(prompts[:2], ar):
        assert _generate(_engine(tw, dw, 3, use_graph=use_graph), [p], 40)[0] == want
    assert _generate(_engine(tw, dw, 3, max_batch=3, rows="narrow", use_graph=use_graph),
                     prompts, 40) == ar
    # wide family (12 verify rows would fit a narrow pass: force it), against its own AR
    eng = _engine(tw, dw, 3, max_batch=3, rows="wide", use_graph=use_graph)
    assert eng.wide
    ar_wide = _generate(_engine(tw, None, 0, max_batch=3, rows="wide", use_graph=use_graph), prompts, 40)
    assert _generate(eng, prompts, 40) == ar_wide


@pytest.mark.parametrize("use_graph", [False, True])
def test_fp8_activations_on_the_wide_family(use_graph):
    """fp8 weights with fp8 activations (W8A8) on the wide family: the logits stay within the
    quantisation error of the bf16-activation pass on the same weights (and are not the same
    numbers), a token's logits do not depend on how the pass was cut, and greedy speculative ==
    autoregressive on a wide engine (verify passes and prompt chunks all W8A8) as on a narrow
    one (prompt chunks only)."""
    tcfg = MODELS["llama"]
    dcfg = OracleConfig(**{**vars(tcfg), "n_layers": 1})
    tw = _product(tcfg, _bf(random_state(tcfg, seed=21)), fp8=True)
    dw = _product(dcfg, _bf(random_state(dcfg, seed=22)), fp8=True)
    toks = torch.randint(0, tcfg.vocab, (60,), generator=torch.Generator().manual_seed(4)).tolist()
    a16 = _logits_via_engine(tw, toks, [40, 20], wide=True)
    a8 = _logits_via_engine(tw, toks, [40, 20], wide=True, fp8_activations=True)
    scale = a16.abs().max().item()
    assert 1e-4 * scale < (a8 - a16).abs().max().item() < 0.15 * scale
    assert (a8.argmax(-1) == a16.argmax(-1)).float().mean().item() > 0.8
    assert torch.equal(a8, _logits_via_engine(tw, toks, [17, 25, 18], wide=True, fp8_activations=True))
    prompts = [[5, 9, 200, 31, 7], list(range(40, 73)), [1, 2, 3]]
    for rows in ("wide", "narrow"):
        kw = dict(max_batch=3, rows=rows, use_graph=use_graph, fp8_activations=True)
        ar = _generate(_engine(tw, None, 0, **kw), prompts, 40)
        eng = _engine(tw, dw, 3, **kw)
        assert eng.target.fp8_activations and eng.draft.fp8_activations
        assert _generate(eng, prompts, 40) == ar
    from vllm_omni_draft_amd.model_runner import DecoderModel
    with pytest.raises(ValueError):
        DecoderModel(_product(tcfg, _bf(random_state(tcfg, seed=21))), 128, 128, fp8_activations=True)


@pytest.mark.parametrize("wide", [False, True])
def test_forward_leaves_the_slice_maxima_of_its_logits(wide):
    """A pass asked for the slice maxima (argmax_parts) gives the argmax of the logits a plain pass
    writes: from the LM head itself on the narrow family (no logits written), through the logits
    on the wide one."""
    from vllm_omni_draft_amd import ops
    from vllm_omni_draft_amd.kv_cache import PagedKVCache
    from vllm_omni_draft_amd.model_runner import DecoderModel
    cfg = MODELS["llama3_rope_gqa4"]
    w = _product(cfg, _bf(random_state(cfg, seed=12)))
    toks = torch.randint(0, cfg.vocab, (30,), generator=torch.Generator().manual_seed(4)).tolist()
    ref = _logits_via_engine(w, toks, [8, 8, 8, 6], wide=wide)
    model = DecoderModel(w, max_ctx=128, split_tokens=128, wide=wide)
    kv = PagedKVCache(cfg, 16, 1, 128, "cuda")
    kv.ensure_capacity(0, len(toks))
    tok = torch.tensor(toks, dtype=torch.int32, device="cuda")
    zero = torch.zeros(1, dtype=torch.int32, device="cuda")
    got, c0 = [], 0
    for q in (8, 8, 8, 6):
        pv = torch.empty(q, ops.ARGMAX_PARTS, dtype=torch.float32, device="cuda")
        pi = torch.empty(q, ops.ARGMAX_PARTS, dtype=torch.int32, device="cuda")
        lg = torch.full((q, cfg.vocab), float("nan"), dtype=torch.float32, device="cuda")
        model.forward(kv, tok, 0, c0, zero, c0, 1, q, lg, 128, argmax_parts=(pv, pi))
        got += ops.argmax_from_partials(pv, pi, q).tolist()
        assert bool(lg.isnan().all()) == (not wide)      # narrow: the head wrote no logits
        c0 += q
    assert got == ref.argmax(-1).tolist()


def test_chunking_is_bit_invariant():
    """Logits of a token do not depend on how many rows shared its pass."""
    ocfg = MODELS["llama"]
    w = _product(ocfg, _bf(random_state(ocfg, seed=1)))
    toks = list(range(3, 27))
    a = _logits_via_engine(w, toks, [8, 8, 8])
    b = _logits_via_engine(w, toks, [1] * 24)
    c = _logits_via_engine(w, toks, [5, 1, 6, 2, 7, 3])
    assert torch.equal(a, b) and torch.equal(a, c)


def _engine(tw, dw, k, **kw):
    from vllm_omni_draft_amd.engine import EngineConfig, SpecEngine
    return SpecEngine(tw, dw, EngineConfig(k=k, max_batch=kw.pop("max_batch", 1), max_seq_len=200,
                                           split_tokens=128, **kw))


def _generate(eng, prompts, n_new):
    outs = [[eng.add_request(i, p)] for i, p in enumerate(prompts)]
    while min(len(o) for o in outs) < n_new:
        for o, new in zip(outs, eng.step()):
            o.extend(new)
    return [o[:n_new] for o in outs]


@pytest.mark.parametrize("use_graph", [False, True])
@pytest.mark.parametrize("k", [1, 3, 5, 8, 15])
def test_greedy_spec_equals_ar_on_gpu(k, use_graph):
    tcfg, dcfg = MODELS["llama"], MODELS["opt"]
    dcfg = OracleConfig(**{**vars(dcfg), "vocab": tcfg.vocab})
    tw = _product(tcfg, _bf(random_state(tcfg, seed=21)))
    dw = _product(dcfg, _bf(random_state(dcfg, seed=22)))
    prompts = [[5, 9, 200, 31, 7], list(range(40, 73))]
    ar = [_generate(_engine(tw, None, 0, use_graph=use_graph), [p], 48)[0] for p in prompts]
    for p, want in zip(prompts, ar):
        eng = _engine(tw, dw, k, use_graph=use_graph)
        got = _generate(eng, [p], 48)[0]
        assert got == want
        m = eng.metrics()
        assert m["draft"] % k == 0 and m["emitted"] - m["accepted"] == m["draft"] // k


@pytest.mark.parametrize("use_graph", [False, True])
def test_overlapped_read_back_gives_the_same_stream(use_graph):
    """step_overlapped enqueues step t+1 before step t is read: same tokens, one call late,
    across a context-bucket change and a page boundary; the batch cannot change in between."""
    tcfg, dcfg = MODELS["llama"], MODELS["opt"]
    dcfg = OracleConfig(**{**vars(dcfg), "vocab": tcfg.vocab})
    tw = _product(tcfg, _bf(random_state(tcfg, seed=21)))
    dw = _product(dcfg, _bf(random_state(dcfg, seed=22)))
    prompts = [[5, 9, 200, 31, 7], list(range(40, 73))]
    want = _generate(_engine(tw, dw, 3, max_batch=2, use_graph=use_graph), prompts, 36)
    eng = _engine(tw, dw, 3, max_batch=2, use_graph=use_graph)
    outs = [[eng.add_request(i, p)] for i, p in enumerate(prompts)]
    assert eng.step_overlapped() is None
    with pytest.raises(RuntimeError):
        eng.step()
    with pytest.raises(RuntimeError):
        eng.remove_request(0)
    while min(len(o) for o in outs) < 36:
        for o, new in zip(outs, eng.step_overlapped()):
            o.extend(new)
    for o, new in zip(outs, eng.flush()):
        o.extend(new)
    assert eng.flush() is None
    assert [o[:36] for o in outs] == want
    # synchronous stepping resumes after the flush, and the metrics saw every step
    more = eng.step()
    assert all(len(t) >= 1 for t in more)
    m = eng.metrics()
    assert m["emitted"] == sum(len(o) - 1 for o in outs) + sum(len(t) for t in more)


def test_draft_equals_target_accepts_all_and_batches():
    """Known answer: with draft == target every proposal is accepted (accept-len == k)."""
    tcfg = MODELS["llama"]
    tw = _product(tcfg, _bf(random_state(tcfg, seed=21)))
    prompts = [[5, 9, 200, 31, 7], [1, 2, 3], list(range(10, 30))]
    ar = [_generate(_engine(tw, None, 0), [p], 40)[0] for p in prompts]
    eng = _engine(tw, tw, 3, max_batch=3)
    got = _generate(eng, prompts, 40)
    assert got == ar
    m = eng.metrics()
    assert m["accepted"] == m["draft"]
    # batched AR equals single AR too
    assert _generate(_engine(tw, None, 0, max_batch=3), prompts, 40) == ar


@pytest.mark.parametrize("temperature", [0.0, 0.8])
def test_prompts_prefilled_together_give_the_streams_of_one_at_a_time(temperature):
    """add_requests carries a chunk of every prompt in each pass (ragged lengths: padding rows;
    one prompt longer than a pass; a one-token prompt).  First tokens, later streams and the
    acceptance counters equal those of add_request called prompt by prompt."""
    tcfg, dcfg = MODELS["llama"], MODELS["opt"]
    dcfg = OracleConfig(**{**vars(dcfg), "vocab": tcfg.vocab})
    tw = _product(tcfg, _bf(random_state(tcfg, seed=21)))
    dw = _product(dcfg, _bf(random_state(dcfg, seed=22)))
    rng = np.random.default_rng(4)
    prompts = [[int(t) for t in rng.integers(0, tcfg.vocab, n)] for n in (5, 33, 1, 150, 17)]
    kw = dict(max_batch=5, temperature=temperature, seed=9)
    one = _engine(tw, dw, 3, **kw)
    want = _generate(one, prompts, 24)
    eng = _engine(tw, dw, 3, **kw)
    firsts = eng.add_requests(prompts)
    assert firsts == [w[0] for w in want]
    outs = [[f] for f in firsts]
    while min(len(o) for o in outs) < 24:
        for o, new in zip(outs, eng.step()):
            o.extend(new)
    assert [o[:24] for o in outs] == want
    assert eng.metrics() == one.metrics()
    # two groups, the second behind the first
    eng = _engine(tw, dw, 3, **kw)
    assert eng.add_requests(prompts[:2]) + eng.add_requests(prompts[2:]) == firsts
    with pytest.raises(ValueError):
        eng.add_requests([[1, 2]])


@pytest.mark.parametrize("use_graph", [False, True])
def test_draft_head_over_a_vocabulary_subset(use_graph):
    """A draft whose LM head covers a subset of the vocabulary: with a permutation of the whole
    vocabulary the proposals (hence the counters) are those of the plain draft; with half of it
    every proposal lies inside the subset; the emitted stream is the target's either way."""
    from vllm_omni_draft_amd import config, weights
    tcfg, dcfg = MODELS["llama"], MODELS["opt"]
    dcfg = OracleConfig(**{**vars(dcfg), "vocab": tcfg.vocab})
    tw = _product(tcfg, _bf(random_state(tcfg, seed=21)))
    dstate = _bf(random_state(dcfg, seed=22))
    dw = _product(dcfg, dstate)
    prompts = [[5, 9, 200, 31, 7], list(range(40, 73))]
    ar = [_generate(_engine(tw, None, 0), [p], 40)[0] for p in prompts]
    plain = _engine(tw, dw, 3, max_batch=2, use_graph=use_graph)
    assert _generate(plain, prompts, 40) == ar
    rng = np.random.default_rng(0)
    perm = [int(t) for t in rng.permutation(tcfg.vocab)]
    half = sorted(perm[:tcfg.vocab // 2])
    for subset, same_counters in ((perm, True), (half, False)):
        dsub = weights.from_state_dict(dw.cfg, dstate, "cuda", head_subset=subset)
        assert dsub.cfg.lm_vocab == len(subset) and dsub.head_map.tolist() == subset
        eng = _engine(tw, dsub, 3, max_batch=2, use_graph=use_graph)
        outs = [[eng.add_request(i, p)] for i, p in enumerate(prompts)]
        while min(len(o) for o in outs) < 40:
            new = eng.step()
            assert set(eng.tok_buf[:2, 2:].flatten().tolist()) <= set(subset)
            for o, n in zip(outs, new):
                o.extend(n)
        assert [o[:40] for o in outs] == ar
        if same_counters:
            assert eng.metrics() == plain.metrics()
    with pytest.raises(ValueError):
        weights.from_state_dict(dw.cfg, dstate, "cuda", head_subset=[1, 2, 3])       # not a multiple of 16
    with pytest.raises(ValueError):
        weights.from_state_dict(dw.cfg, dstate, "cuda", head_subset=[0] * 16)        # duplicates
    with pytest.raises(ValueError):
        _engine(tw, weights.from_state_dict(dw.cfg, dstate, "cuda", head_subset=half), 3, temperature=0.7)
    with pytest.raises(ValueError):
        _engine(weights.from_state_dict(tw.cfg, _bf(random_state(tcfg, seed=21)), "cuda", head_subset=half),
                dw, 3)


@pytest.mark.parametrize("temperature", [0.0, 0.8])
def test_prompts_prefilled_in_pieces_between_steps(temperature):
    """prefill_begin / prefill_advance (a scheduler with chunked prefill): prompts enter piece by
    piece while the running sequences keep stepping, two prompts at once finishing in the other
    order than they began, a whole prompt admitted in between, a neighbour leaving; every stream
    equals the one the request gets alone."""
    tcfg, dcfg = MODELS["llama"], MODELS["opt"]
    dcfg = OracleConfig(**{**vars(dcfg), "vocab": tcfg.vocab})
    tw = _product(tcfg, _bf(random_state(tcfg, seed=21)))
    dw = _product(dcfg, _bf(random_state(dcfg, seed=22)))
    rng = np.random.default_rng(8)
    P = [[int(t) for t in rng.integers(0, tcfg.vocab, n)] for n in (9, 70, 33, 12, 1)]
    kw = dict(temperature=temperature)

    def alone(i, n=26):
        eng = _engine(tw, dw, 3, max_batch=1, **kw)
        out = [eng.add_request(0, P[i], seed=100 + i)]
        while len(out) < n:
            out += eng.step()[0]
        return out[:n]

    want = [alone(i) for i in range(len(P))]
    eng = _engine(tw, dw, 3, max_batch=4, **kw)
    outs = {0: [eng.add_request(0, P[0], seed=100)]}
    order = [0]                                            # request at each active slot

    def step():
        for i, new in zip(order, eng.step()):
            outs[i].extend(new)

    h1 = eng.prefill_begin(P[1], seed=101)                 # 70 tokens in pieces of 32, 30, 8
    h2 = eng.prefill_begin(P[2], seed=102)                 # 33 tokens in pieces of 32, 1
    assert eng.prefill_advance(h1, 32) is None
    step()
    assert eng.prefill_advance(h2, 32) is None
    outs[3] = [eng.add_request(eng.active, P[3], seed=103)]   # a whole prompt in between
    order.append(3)
    step()
    assert eng.prefill_advance(h1, 30) is None
    first = eng.prefill_advance(h2, 1)                     # the later prompt finishes first
    assert first is not None
    outs[2] = [first]
    order.append(2)
    step()
    eng.remove_request(0)                                  # request 0 leaves: the last active moves in
    order[0] = order.pop()
    step()
    outs[1] = [eng.prefill_advance(h1, 8)]
    order.append(1)
    with pytest.raises(KeyError):
        eng.prefill_advance(h1, 1)
    h4 = eng.prefill_begin(P[4], seed=104)                 # a one-token prompt: one piece
    outs[4] = [eng.prefill_advance(h4, 1)]
    order.append(4)
    with pytest.raises(ValueError):
        eng.prefill_begin(P[0])                            # the batch is full
    while min(len(outs[i]) for i in (1, 2, 3, 4)) < 26:
        step()
    for i in (1, 2, 3, 4):
        assert outs[i][:26] == want[i], i
    assert outs[0] == want[0][:len(outs[0])]
    # abort frees the slot
    eng.remove_request(0)
    h = eng.prefill_begin(P[1])
    eng.prefill_advance(h, 10)
    eng.prefill_abort(h)
    assert eng.prefill_begin(P[0]) is not None


@pytest.mark.parametrize("use_graph", [False, True])
def test_wide_batch_spec_equals_ar(use_graph):
    """8 sequences x (k+1 = 6) rows = 48 rows per verify pass: the wide kernel family.  The
    speculative stream equals the autoregressive one of an engine on the same family."""
    tcfg, dcfg = MODELS["llama"], MODELS["opt"]
    dcfg = OracleConfig(**{**vars(dcfg), "vocab": tcfg.vocab})
    tw = _product(tcfg, _bf(random_state(tcfg, seed=21)))
    dw = _product(dcfg, _bf(random_state(dcfg, seed=22)))
    g = torch.Generator().manual_seed(9)
    prompts = [torch.randint(0, tcfg.vocab, (int(n),), generator=g).tolist()
               for n in torch.randint(3, 40, (8,), generator=g)]
    ar_eng = _engine(tw, None, 0, max_batch=8, rows="wide", use_graph=use_graph)
    assert ar_eng.wide
    ar = _generate(ar_eng, prompts, 40)
    eng = _engine(tw, dw, 5, max_batch=8, use_graph=use_graph)
    assert eng.wide and eng.max_rows == 256
    got = _generate(eng, prompts, 40)
    assert got == ar
    # draft == target on the wide family: every proposal accepted
    eng2 = _engine(tw, tw, 3, max_batch=8, rows="wide", use_graph=use_graph)
    assert _generate(eng2, prompts, 40) == ar
    m = eng2.metrics()
    assert m["accepted"] == m["draft"]


def test_wide_batch_spec_equals_ar_with_seven_head_groups():
    """Qwen2-like heads (7 query heads per KV head, QKV bias): a verify pass has 42 query rows
    per KV head (two-M-tile attention kernel), an autoregressive step 7 (one-tile kernel), and
    the streams must still be equal."""
    tcfg = MODELS["qwen2"]
    dcfg = OracleConfig(**{**vars(MODELS["llama"]), "vocab": tcfg.vocab})
    tw = _product(tcfg, _bf(random_state(tcfg, seed=31, qkv_bias=True)), qkv_bias=True)
    dw = _product(dcfg, _bf(random_state(dcfg, seed=32)))
    g = torch.Generator().manual_seed(10)
    prompts = [torch.randint(0, tcfg.vocab, (int(n),), generator=g).tolist()
               for n in torch.randint(3, 150, (6,), generator=g)]
    ar = _generate(_engine(tw, None, 0, max_batch=6, rows="wide"), prompts, 30)
    eng = _engine(tw, dw, 5, max_batch=6)
    assert eng.wide
    assert _generate(eng, prompts, 30) == ar
    # and alone in the narrow family (6 verify rows), against the narrow autoregressive engine
    ar1 = _generate(_engine(tw, None, 0), prompts[:1], 30)
    assert _generate(_engine(tw, dw, 5), prompts[:1], 30) == ar1


def test_synthetic_acceptance_control():
    tcfg, dcfg = MODELS["llama"], MODELS["llama"]
    tw = _product(tcfg, _bf(random_state(tcfg, seed=21)))
    dw = _product(dcfg, _bf(random_state(dcfg, seed=99)))
    prompt = [7, 7, 8, 100, 3, 2]
    ar = _generate(_engine(tw, None, 0), [prompt], 64)[0]
    truth = torch.zeros(1, 200, dtype=torch.int32)
    truth[0, :len(prompt)] = torch.tensor(prompt)
    truth[0, len(prompt):len(prompt) + 64] = torch.tensor(ar)
    from vllm_omni_draft_amd.engine import acceptance_pattern
    for a in (0.0, 0.7, 1.0):
        eng = _engine(tw, dw, 4)
        eng.set_synthetic_acceptance(truth.cuda(), a, seed=5, pattern_len=8)
        got = _generate(eng, [prompt], 56)[0]
        assert got == ar[:56]
        m = eng.metrics()
        steps = m["draft"] // 4
        pat = acceptance_pattern(a, 4, 8, rotate=5)
        # deterministic: exactly the schedule's accepted counts, step by step
        assert m["accepted"] == sum(pat[t % 8] for t in range(steps)), (a, m)
        assert m["emitted"] == m["accepted"] + steps
    # an explicit schedule
    eng = _engine(tw, dw, 4)
    eng.set_synthetic_acceptance(truth.cuda(), pattern=[4, 0, 2])
    assert _generate(eng, [prompt], 40)[0] == ar[:40]
    m = eng.metrics()
    assert m["accepted"] == sum([4, 0, 2][t % 3] for t in range(m["draft"] // 4))


def test_golden_e2e_stream_matches_oracle():
    """Token streams of the tiny golden models: HIP engine vs the committed oracle
    output, where the oracle's top-2 logit margin is wide enough for bf16."""
    with open(os.path.join(GOLDEN, "e2e_tiny.json")) as f:
        g = json.load(f)
    tcfg, dcfg = OracleConfig(**g["target"]), OracleConfig(**g["draft"])
    ts, ds = _bf(random_state(tcfg, g["target_seed"])), _bf(random_state(dcfg, g["draft_seed"]))
    tw, dw = _product(tcfg, ts), _product(dcfg, ds)
    lm = OracleLM(tcfg, ts, bf16_acts=True)
    n_tokens = n_match = 0
    for case in g["cases"]:
        eng = _engine(tw, dw, g["k"])
        got = _generate(eng, [case["prompt"]], len(case["tokens"]))[0]
        # compare up to the first position where the oracle itself is near a tie
        seq = list(case["prompt"])
        cache = lm.new_cache()
        logits = lm.forward(seq, 0, cache)[-1]
        for i, want in enumerate(case["tokens"]):
            if got[i] != want:
                # a disagreement is only legitimate at a near-tie of the oracle's logits
                top2 = torch.topk(logits, 2).values
                margin = (top2[0] - top2[1]).item() / logits.abs().max().item()
                assert margin < 0.03, (i, got[i], want, margin)
                break
            n_match += 1
            logits = lm.forward([want], len(seq), cache)[-1]
            seq.append(want)
        n_tokens += len(case["tokens"])
    assert n_match >= 0.6 * n_tokens, (n_match, n_tokens)


def _markov_pair(V=32, seed=5):
    """A target and a draft whose next-token distribution depends on the previous token only
    (output and down projections zeroed: the residual stream stays the token embedding), so
    that EVERY emitted token is a draw from a known row of a V x V matrix, whatever its
    history; the two models have different matrices, so proposals do get rejected."""
    cfg = OracleConfig(arch="llama", n_layers=1, hidden=64, n_heads=1, n_kv_heads=1, head_dim=64,
                       ffn=64, vocab=V, max_pos=1024)
    states = []
    for sd in (seed, seed + 1):
        st = random_state(cfg, seed=sd, std=0.3)
        for k in st:
            if k.endswith("o_proj.weight") or k.endswith("down_proj.weight"):
                st[k] = torch.zeros_like(st[k])
        states.append(st)
    # the draft: the target pulled towards another random model (a draft that agrees with its
    # target about two times in three)
    states[1] = {k: 0.75 * states[0][k] + 0.25 * states[1][k] for k in states[0]}
    return cfg, [_bf(st) for st in states]


def _transition_matrix(cfg, state, temperature):
    lm = OracleLM(cfg, state, bf16_acts=True)
    rows = [torch.softmax(lm.forward([t], 0, lm.new_cache())[-1].double() / temperature, dim=0)
            for t in range(cfg.vocab)]
    return torch.stack(rows).numpy()


# Stated tolerance of the sampling path: over
# N >= 2e4 emitted tokens, Pearson's chi-square of the (previous, next) token counts against
# the oracle's conditional distribution, over the cells with an expected count >= 5, has a
# p-value above 1e-5, and the count-weighted KL divergence of the empirical conditional
# distribution from the oracle's stays below 0.05 nats (its expectation under an exact
# sampler is (V - 1) V / 2N = 0.025 at V = 32, N = 2e4).
SAMPLING_P_VALUE = 1e-5
SAMPLING_KL_NATS = 0.05


def _check_transitions(counts, P, min_total=20000):
    from scipy import stats
    n_prev = counts.sum(axis=1, keepdims=True)
    N = counts.sum()
    assert N >= min_total, N
    exp = n_prev * P
    mask = exp >= 5
    chi2 = ((counts - exp) ** 2 / np.where(mask, exp, 1.0))[mask].sum()
    dof = int(mask.sum()) - int((mask.sum(axis=1) > 0).sum())
    pval = float(stats.chi2.sf(chi2, dof))
    with np.errstate(divide="ignore", invalid="ignore"):
        emp = counts / np.maximum(n_prev, 1)
        kl = np.where(counts > 0, counts * np.log(emp / P), 0.0).sum() / N
    assert pval > SAMPLING_P_VALUE, (chi2, dof, pval)
    assert kl < SAMPLING_KL_NATS, kl
    return pval, kl


@pytest.mark.parametrize("B,use_graph,method", [(4, True, "rejection_sampler"),
                                                (4, False, "rejection_sampler"),
                                                (32, True, "rejection_sampler")])
def test_sampled_tokens_follow_the_oracle_target_distribution(B, use_graph, method):
    """Engine-level sampling test at T = 1: k draft passes with on-device draws, rejection
    sampling with device-drawn uniforms, residual and bonus draws, narrow (B=4) and wide
    (B=32) kernel families, with and without the step graph.  Every emitted token must be
    distributed as the ORACLE's target distribution given its predecessor; the draft's own
    distribution (a different matrix) must not leak into the output."""
    cfg, (ts, ds) = _markov_pair()
    P = _transition_matrix(cfg, ts, 1.0)
    Q = _transition_matrix(cfg, ds, 1.0)
    assert np.abs(P - Q).sum(axis=1).mean() > 0.3          # the two models do disagree
    tw, dw = _product(cfg, ts), _product(cfg, ds)
    from vllm_omni_draft_amd.engine import EngineConfig, SpecEngine
    eng = SpecEngine(tw, dw, EngineConfig(k=3, max_batch=B, max_seq_len=900, split_tokens=128,
                                          temperature=1.0, seed=11, use_graph=use_graph,
                                          acceptance_method=method))
    counts = np.zeros((cfg.vocab, cfg.vocab))
    rounds = 0
    while counts.sum() < 21000:
        eng.reset()
        seqs = [[(7 * b + rounds) % cfg.vocab] for b in range(B)]      # one-token prompts
        for b in range(B):
            seqs[b].append(eng.add_request(b, seqs[b]))
        for _ in range(280 if B == 4 else 40):
            for b, new in enumerate(eng.step()):
                assert 1 <= len(new) <= 4 and all(0 <= t < cfg.vocab for t in new)
                seqs[b].extend(new)
        for sq in seqs:
            np.add.at(counts, (sq[:-1], sq[1:]), 1)
        rounds += 1
    m = eng.metrics()
    assert 0.2 < m["accepted"] / m["draft"] < 0.9          # both accept and reject paths ran
    _check_transitions(counts, P)
    # sensitivity: the same counts do NOT pass against the draft's distribution
    with pytest.raises(AssertionError):
        _check_transitions(counts, Q)


@pytest.mark.parametrize("use_graph", [False, True])
def test_a_seeded_request_samples_the_same_tokens_alone_and_in_a_batch(use_graph):
    """Every sequence draws from its own stream (its seed, its own step counter): the tokens a
    seeded request samples do not depend on its batch mates, its slot, how it was admitted
    (alone or in a batched prefill) or a move to another slot when a neighbour leaves."""
    tcfg, dcfg = MODELS["llama"], MODELS["opt"]
    dcfg = OracleConfig(**{**vars(dcfg), "vocab": tcfg.vocab})
    tw = _product(tcfg, _bf(random_state(tcfg, seed=21)))
    dw = _product(dcfg, _bf(random_state(dcfg, seed=22)))
    pa, pb, pc = [5, 9, 200, 31, 7], list(range(40, 60)), [3, 1, 4, 1, 5, 9, 2, 6]
    kw = dict(temperature=0.8, use_graph=use_graph)

    def alone(prompt, seed, n=30):
        eng = _engine(tw, dw, 3, max_batch=1, **kw)
        out = [eng.add_request(0, prompt, seed=seed)]
        while len(out) < n:
            out += eng.step()[0]
        return out[:n]

    a5, b7 = alone(pa, 5), alone(pb, 7)
    assert alone(pa, 5) == a5 and alone(pa, 6) != a5              # reproducible, and the seed matters
    eng = _engine(tw, dw, 3, max_batch=3, **kw)
    outs = [[t] for t in eng.add_requests([pc, pa, pb], seeds=[None, 5, 7])]     # slots 0, 1, 2
    for _ in range(4):
        for o, new in zip(outs, eng.step()):
            o.extend(new)
    eng.remove_request(0)                                         # pb moves from slot 2 to slot 0
    outs = [outs[2], outs[1]]
    while min(len(o) for o in outs) < 30:
        for o, new in zip(outs, eng.step()):
            o.extend(new)
    assert outs[1][:30] == a5 and outs[0][:30] == b7


@pytest.mark.parametrize("use_graph", [False, True])
def test_requests_with_their_own_temperatures_share_a_batch(use_graph):
    """A sampling engine serves every request at its own temperature: a greedy request (0) gets
    the greedy autoregressive stream, seeded requests at 0.8 and 1.3 get the streams they get
    alone on an engine of that temperature."""
    tcfg, dcfg = MODELS["llama"], MODELS["opt"]
    dcfg = OracleConfig(**{**vars(dcfg), "vocab": tcfg.vocab})
    tw = _product(tcfg, _bf(random_state(tcfg, seed=21)))
    dw = _product(dcfg, _bf(random_state(dcfg, seed=22)))
    pa, pb, pc = [5, 9, 200, 31, 7], list(range(40, 60)), [3, 1, 4, 1, 5, 9, 2, 6]
    greedy = _generate(_engine(tw, None, 0), [pa], 30)[0]

    def alone(prompt, seed, T):
        eng = _engine(tw, dw, 3, max_batch=1, temperature=T, use_graph=use_graph)
        out = [eng.add_request(0, prompt, seed=seed)]
        while len(out) < 30:
            out += eng.step()[0]
        return out[:30]

    b08, c13 = alone(pb, 7, 0.8), alone(pc, 9, 1.3)
    assert b08 != alone(pb, 7, 1.3)                                  # the temperature matters
    eng = _engine(tw, dw, 3, max_batch=3, temperature=0.8, use_graph=use_graph)
    outs = [[eng.add_request(0, pa, temperature=0.0)], [eng.add_request(1, pb, seed=7)],
            [eng.add_request(2, pc, seed=9, temperature=1.3)]]
    while min(len(o) for o in outs) < 30:
        for o, new in zip(outs, eng.step()):
            o.extend(new)
    assert outs[0][:30] == greedy and outs[1][:30] == b08 and outs[2][:30] == c13
    with pytest.raises(ValueError):
        _engine(tw, dw, 3).add_request(0, pa, temperature=0.7)       # a greedy engine cannot sample
    with pytest.raises(ValueError):
        eng.add_requests([pa], temperatures=[-1.0])


def test_top_k_top_p_requests_follow_the_filtered_target_distribution():
    """Requests with top_k / top_p: every emitted token is a draw from the oracle's FILTERED target
    distribution given its predecessor (the draft proposes from its whole distribution, rejection
    sampling corrects it), tokens outside the filter never appear, and top_k = 1 on a sampling
    engine is greedy decoding."""
    from oracle import sampling as osamp
    cfg, (ts, ds) = _markov_pair()
    P = _transition_matrix(cfg, ts, 1.0)
    Pf = np.stack([osamp.filter_probs(row, top_k=6, top_p=0.9) for row in P])
    assert ((Pf > 0).sum(axis=1) <= 6).all() and (Pf > 0).sum() < 0.25 * P.size
    tw, dw = _product(cfg, ts), _product(cfg, ds)
    from vllm_omni_draft_amd.engine import EngineConfig, SpecEngine
    B = 4
    eng = SpecEngine(tw, dw, EngineConfig(k=3, max_batch=B, max_seq_len=900, split_tokens=128,
                                          temperature=1.0, seed=3))
    counts = np.zeros((cfg.vocab, cfg.vocab))
    rounds = 0
    while counts.sum() < 21000:
        eng.reset()
        seqs = [[(7 * b + rounds) % cfg.vocab] for b in range(B)]
        firsts = eng.add_requests(seqs, filters=[dict(top_k=6, top_p=0.9)] * B)
        for sq, f in zip(seqs, firsts):
            sq.append(f)
        for _ in range(280):
            for b, new in enumerate(eng.step()):
                seqs[b].extend(new)
        for sq in seqs:
            np.add.at(counts, (sq[:-1], sq[1:]), 1)
        rounds += 1
    assert counts[Pf == 0].sum() == 0                       # nothing outside the filter, ever
    _check_transitions(counts, Pf)
    with pytest.raises(AssertionError):
        _check_transitions(counts, P)                       # and it is not the unfiltered one
    # top_k = 1 (and a nucleus smaller than the most likely token) leave only that token
    tcfg = MODELS["llama"]
    tw2 = _product(tcfg, _bf(random_state(tcfg, seed=21)))
    prompt = [5, 9, 200, 31, 7]
    greedy = _generate(_engine(tw2, None, 0), [prompt], 30)[0]
    for flt in (dict(top_k=1), dict(top_p=1e-4)):
        e2 = _engine(tw2, tw2, 3, temperature=0.9)
        out = [e2.add_request(0, prompt, seed=1, **flt)]
        while len(out) < 30:
            out += e2.step()[0]
        assert out[:30] == greedy, flt


@pytest.mark.parametrize("use_graph", [False, True])
def test_typical_acceptance_engine_known_answers(use_graph):
    """Typical acceptance at its two extremes.  A threshold nothing can exceed rejects every
    proposal and emits the target's most likely token: the stream after the (sampled) first
    token is the greedy continuation.  A zero threshold lets every proposal stand."""
    tcfg, dcfg = MODELS["llama"], MODELS["opt"]
    dcfg = OracleConfig(**{**vars(dcfg), "vocab": tcfg.vocab})
    tw = _product(tcfg, _bf(random_state(tcfg, seed=21)))
    dw = _product(dcfg, _bf(random_state(dcfg, seed=22)))
    prompt = [5, 9, 200, 31, 7]
    kw = dict(temperature=1.0, seed=3, use_graph=use_graph,
              acceptance_method="typical_acceptance_sampler")
    never = _engine(tw, dw, 3, posterior_threshold=2.0, posterior_alpha=1e9, **kw)
    toks = _generate(never, [prompt], 24)[0]
    m = never.metrics()
    assert m["accepted"] == 0 and m["draft"] > 0
    greedy = _generate(_engine(tw, None, 0, use_graph=use_graph), [prompt + toks[:1]], 23)[0]
    assert toks[1:] == greedy
    always = _engine(tw, dw, 3, posterior_threshold=0.0, posterior_alpha=0.0, **kw)
    _generate(always, [prompt], 24)
    m = always.metrics()
    assert m["draft"] > 0 and m["accepted"] == m["draft"]
    with pytest.raises(ValueError):
        _engine(tw, dw, 3, acceptance_method="nonsense")


@pytest.mark.parametrize("use_graph", [False, True])
def test_prompt_lookup_proposer_equals_ar_and_accepts_repeats(use_graph):
    """No draft model: n-gram proposals from the sequence's own history.  Greedy stream ==
    autoregressive stream; on a sequence that repeats, proposals are accepted."""
    tcfg = MODELS["llama"]
    tw = _product(tcfg, _bf(random_state(tcfg, seed=21)))
    prompts = [[5, 9, 200, 31, 7, 5, 9, 200], list(range(40, 60)) * 2]
    ar = [_generate(_engine(tw, None, 0, use_graph=use_graph), [p], 60)[0] for p in prompts]
    eng = _engine(tw, None, 4, max_batch=2, ngram_max=3, ngram_min=1, use_graph=use_graph)
    assert eng.ngram and eng.draft is None
    got = _generate(eng, prompts, 60)
    assert got == ar
    m = eng.metrics()
    # a tiny random model falls into a loop quickly: the lookup then predicts it
    assert m["draft"] > 0 and m["accepted"] > 0
    # long lookups (k = 12: 13 verify rows): once the model loops, whole runs are accepted
    eng = _engine(tw, None, 12, max_batch=1, ngram_max=3, ngram_min=1, use_graph=use_graph)
    assert _generate(eng, prompts[:1], 60) == ar[:1]
    m12 = eng.metrics()
    assert m12["accepted"] > 0 and (m12["emitted"] - m12["accepted"]) < 60     # fewer steps than tokens
    with pytest.raises(ValueError):
        _engine(tw, None, 16, ngram_max=3)                                     # beyond the fused walk
    # sampling mode runs with the one-hot draft distribution of the lookup
    eng = _engine(tw, None, 4, max_batch=2, ngram_max=3, temperature=1.0, seed=1, use_graph=use_graph)
    out = _generate(eng, prompts, 30)
    assert all(0 <= t < tcfg.vocab for o in out for t in o)


@pytest.mark.parametrize("use_graph", [False, True])
def test_disable_by_batch_size_keeps_streams_and_draft_cache(use_graph):
    """Above the threshold a step emits one target token per sequence and proposes nothing;
    when a sequence leaves, speculation resumes with an intact draft cache (draft == target:
    every proposal accepted again)."""
    tcfg = MODELS["llama"]
    tw = _product(tcfg, _bf(random_state(tcfg, seed=21)))
    prompts = [[5, 9, 200, 31, 7], [1, 2, 3], list(range(10, 30))]
    ar = [_generate(_engine(tw, None, 0, use_graph=use_graph), [p], 40)[0] for p in prompts]
    eng = _engine(tw, tw, 3, max_batch=3, disable_by_batch_size=2, use_graph=use_graph)
    outs = [[eng.add_request(i, p)] for i, p in enumerate(prompts)]
    for _ in range(6):                                   # 3 running > 2: no speculation
        for o, new in zip(outs, eng.step()):
            assert len(new) == 1
            o.extend(new)
    assert eng.metrics()["draft"] == 0
    eng.remove_request(2)                                # 2 running: speculate again
    done = outs.pop(2)
    while min(len(o) for o in outs) < 40:
        for o, new in zip(outs, eng.step()):
            o.extend(new)
    m = eng.metrics()
    assert m["draft"] > 0 and m["accepted"] == m["draft"]
    assert [o[:40] for o in outs] == ar[:2] and done == ar[2][:len(done)]


@pytest.mark.parametrize("use_graph", [False, True])
@pytest.mark.parametrize("rows", ["narrow", "wide"])
def test_the_speculation_policy_changes_no_greedy_stream(use_graph, rows):
    """spec_policy="auto" with a probe every 3 steps: the engine keeps switching between
    speculative steps and plain target steps that leave the draft out; the draft's cache is
    caught up by a wide pass in between (also for a one-token prompt and a sequence admitted
    mid-way).  Every stream is the autoregressive one on both kernel families, both kinds of step
    ran, and with draft == target nearly every proposal is accepted (the draft's cache is intact
    after every catch-up; the wide pass rounds differently, so not necessarily all)."""
    from tests.cpu_backend.case_cpu_backend import _policy_run
    from vllm_omni_draft_amd.engine import EngineConfig, SpecEngine
    tcfg = MODELS["llama"]
    tw = _product(tcfg, _bf(random_state(tcfg, seed=21)))
    rng = torch.Generator().manual_seed(5)
    prompts = [torch.randint(0, tcfg.vocab, (n,), generator=rng).tolist() for n in (9, 1, 30, 3)]
    want, _, _ = _policy_run(SpecEngine, EngineConfig, tw, None, prompts, 40, k=0, rows=rows,
                             use_graph=use_graph)
    got, modes, eng = _policy_run(SpecEngine, EngineConfig, tw, tw, prompts, 40, k=3, rows=rows,
                                  use_graph=use_graph, spec_policy="auto", policy_probe_every=3,
                                  policy_probe_steps=2)
    assert got[0] == want[0] and got[2] == want[2] and got[3][:20] == want[3][:20]
    assert got[1][:20] == want[1][:20]
    assert any(modes) and not all(modes)
    m = eng.metrics()
    assert m["draft"] > 0 and m["accepted"] >= 0.85 * m["draft"], m


@pytest.mark.parametrize("use_graph", [False, True])
def test_narrow_draft_sub_batches_on_parallel_streams(use_graph):
    """20 sequences: the narrow draft's passes run as sub-batches of 16 rows on parallel
    streams (forked and joined inside the step).  The stream equals the autoregressive one and
    the one with the sub-batches in sequence, and the proposals are the same either way."""
    tcfg, dcfg = MODELS["llama"], MODELS["opt"]
    dcfg = OracleConfig(**{**vars(dcfg), "vocab": tcfg.vocab})
    tw = _product(tcfg, _bf(random_state(tcfg, seed=21)))
    dw = _product(dcfg, _bf(random_state(dcfg, seed=22)))
    g = torch.Generator().manual_seed(5)
    prompts = [torch.randint(0, tcfg.vocab, (int(n),), generator=g).tolist()
               for n in torch.randint(3, 40, (20,), generator=g)]
    kw = dict(max_batch=20, use_graph=use_graph, rows="wide")
    ar = _generate(_engine(tw, None, 0, **kw), prompts, 30)
    results = []
    for lanes in (4, 1):
        eng = _engine(tw, dw, 3, draft_rows="narrow", draft_lanes=lanes, **kw)
        assert len(eng._lanes) == (2 if lanes > 1 else 0)     # 40 rows of the first pass: 3 lanes
        assert _generate(eng, prompts, 30) == ar
        results.append(eng.metrics())
    assert results[0] == results[1]
    auto = _engine(tw, dw, 3, **kw)
    assert not auto.draft.wide and auto._lanes      # the default picks this path at 20 sequences


def test_config1_opt125m_target_and_draft_known_answer():
    """BASELINE.json configs[0] on the GPU library: OPT-125m shapes, the same weights as target
    and draft, k = 3, greedy, batch 8.  Known answer: every proposal is accepted (accept-len ==
    k exactly, 4 tokens per step and sequence); the streams are the oracle's greedy
    autoregressive streams up to its first near-tie."""
    oc = OracleConfig(arch="opt", n_layers=12, hidden=768, n_heads=12, n_kv_heads=12, head_dim=64,
                      ffn=3072, vocab=50272, max_pos=256, tie_embeddings=True)
    st = _bf(random_state(oc, seed=125))
    w = _product(oc, st)
    g = torch.Generator().manual_seed(1234)
    prompts = [torch.randint(0, oc.vocab, (int(n),), generator=g).tolist()
               for n in torch.randint(4, 24, (8,), generator=g)]
    from vllm_omni_draft_amd.engine import EngineConfig, SpecEngine
    eng = SpecEngine(w, w, EngineConfig(k=3, max_batch=8, max_seq_len=96, split_tokens=128))
    outs = [[eng.add_request(i, p)] for i, p in enumerate(prompts)]
    for _ in range(4):
        for o, new in zip(outs, eng.step()):
            assert len(new) == 4, new
            o.extend(new)
    m = eng.metrics()
    assert m["accepted"] == m["draft"] == 8 * 4 * 3 and m["emitted"] == 8 * 4 * 4
    # the same streams from an autoregressive engine of the same kernel family
    ar = SpecEngine(w, None, EngineConfig(k=0, max_batch=8, max_seq_len=96, split_tokens=128,
                                          rows="wide" if eng.wide else "narrow"))
    assert _generate(ar, prompts, 17) == [o[:17] for o in outs]
    lm = OracleLM(oc, st, bf16_acts=True)
    n_match = n_all = 0
    for p, got in list(zip(prompts, outs))[:3]:         # the oracle is slow at this vocabulary
        seq, cache = list(p), lm.new_cache()
        logits = lm.forward(seq, 0, cache)[-1]
        for i in range(len(got)):
            want = int(torch.argmax(logits))
            if got[i] != want:                          # only legitimate at a near-tie of the oracle
                top2 = torch.topk(logits, 2).values
                assert (top2[0] - top2[1]).item() / logits.abs().max().item() < 0.03, i
                break
            n_match += 1
            logits = lm.forward([want], len(seq), cache)[-1]
            seq.append(want)
        n_all += len(got)
    assert n_match >= 0.6 * n_all, (n_match, n_all)
