"""Log-probabilities of emitted tokens (ModelRunnerOutput.logprobs): the engine reads them from the
target's logits rows of the step, so a speculative stream carries the values the autoregressive
stream does; through the worker they arrive as one row per emitted token."""
import numpy as np
import pytest
import torch

from oracle.models import OracleConfig, random_state
from tests.test_gpu_model import MODELS, _bf, _engine, _product
from tests.test_gpu_tree import _repetitive_prompts

pytestmark = pytest.mark.gpu


def _pair():
    tcfg = MODELS["llama"]
    dcfg = OracleConfig(**{**vars(MODELS["opt"]), "vocab": tcfg.vocab})
    return (_product(tcfg, _bf(random_state(tcfg, seed=21))),
            _product(dcfg, _bf(random_state(dcfg, seed=22))))


def _run(eng, prompts, n_new, top, seeds=None, together=False):
    """Streams and their log-probability entries, per prompt."""
    eng.logprobs_top = top
    toks, lps = [], []
    if together:
        firsts = eng.add_requests(prompts, seeds=seeds) if seeds else eng.add_requests(prompts)
        for i, t in enumerate(firsts):
            toks.append([t])
            lps.append(list(eng.last_logprobs[i]))
    else:
        for i, p in enumerate(prompts):
            kw = {"seed": seeds[i]} if seeds else {}
            toks.append([eng.add_request(i, p, **kw)])
            lps.append(list(eng.last_logprobs[0]))
    while min(len(t) for t in toks) < n_new:
        out = eng.step()
        for b, new in enumerate(out):
            toks[b].extend(new)
            assert len(eng.last_logprobs[b]) == len(new)
            lps[b].extend(eng.last_logprobs[b])
    return [t[:n_new] for t in toks], [e[:n_new] for e in lps]


def _check_greedy_entries(toks, lps, top):
    for ts, es in zip(toks, lps):
        for t, (ids, vals, rank) in zip(ts, es):
            assert len(ids) == len(vals) == top + 1
            assert ids[0] == t and rank == 1           # greedy: the emitted token is its row's maximum
            if top:
                assert ids[1] == t and vals[1] == vals[0]
                assert all(vals[i] >= vals[i + 1] for i in range(1, top))
            assert vals[0] <= 0.0


@pytest.mark.parametrize("use_graph", [False, True])
def test_speculative_stream_carries_the_autoregressive_logprobs(use_graph):
    """Greedy: the rows a verify pass scores are bit-identical to the autoregressive passes' rows,
    so every emitted token has the same log-probability entry in both engines; each entry's top
    token is the emitted one (a wrong row would not have rank 1)."""
    tw, dw = _pair()
    prompts = [[5, 9, 200, 31, 7], list(range(40, 73)), [3, 3, 3]]
    ar_t, ar_l = _run(_engine(tw, None, 0, max_batch=3, use_graph=use_graph), prompts, 30, 3)
    _check_greedy_entries(ar_t, ar_l, 3)
    sp_t, sp_l = _run(_engine(tw, dw, 3, max_batch=3, use_graph=use_graph), prompts, 30, 3, together=True)
    assert sp_t == ar_t and sp_l == ar_l
    # against torch on the engine's own logits: the first token of a fresh request
    eng = _engine(tw, dw, 3, use_graph=use_graph)
    eng.logprobs_top = 4
    first = eng.add_request(0, prompts[1])
    ids, vals, rank = eng.last_logprobs[0][0]
    ref = torch.log_softmax(eng.t_logits[0].double(), dim=-1)
    top = torch.topk(ref, 4)
    assert ids == [first] + top.indices.tolist()
    np.testing.assert_allclose(vals, [ref[first].item()] + top.values.tolist(), atol=2e-5)
    # switched off again: nothing is computed
    eng.logprobs_top = None
    eng.last_logprobs = None
    eng.step()
    assert eng.last_logprobs is None


def test_logprobs_of_a_step_that_does_not_speculate_and_of_tree_steps():
    tw, dw = _pair()
    prompts = _repetitive_prompts(MODELS["llama"].vocab)
    ar_t, ar_l = _run(_engine(tw, None, 0, max_batch=3), prompts, 30, 1)
    # above disable_by_batch_size a step is one target token per sequence
    ns_t, ns_l = _run(_engine(tw, dw, 3, max_batch=3, disable_by_batch_size=2), prompts, 30, 1)
    assert ns_t == ar_t and ns_l == ar_l
    # token trees: an emitted token's row is the root's or its accepted parent's
    tr_t, tr_l = _run(_engine(tw, None, 3, max_batch=3, ngram_max=3, ngram_min=1, tree_candidates=4),
                      prompts, 30, 1)
    assert tr_t == ar_t
    _check_greedy_entries(tr_t, tr_l, 1)
    for a, b in zip(ar_l, tr_l):      # tree rows round like chain rows up to one bf16 ulp of attention
        for (ia, va, ra), (ib, vb, rb) in zip(a, b):
            assert ia[0] == ib[0] and ra == rb and abs(va[0] - vb[0]) < 0.05
    # prompt lookup (chains with ragged proposal lengths)
    ng_t, ng_l = _run(_engine(tw, None, 3, max_batch=3, ngram_max=3, ngram_min=1), prompts, 30, 1)
    assert ng_t == ar_t and ng_l == ar_l


def test_sampled_tokens_report_raw_logprobs_and_ranks():
    """Sampling at a temperature: entries come from the raw logits (upstream's default), so the
    emitted token need not be the top one; a seeded request reports the same entries alone as in
    a batch; an overlapped or multi-step read-back is refused."""
    tw, dw = _pair()
    pa, pb = [5, 9, 200, 31, 7], list(range(40, 60))
    kw = dict(temperature=0.9)
    a_t, a_l = _run(_engine(tw, dw, 3, **kw), [pa], 24, 2, seeds=[5])
    both_t, both_l = _run(_engine(tw, dw, 3, max_batch=2, **kw), [pb, pa], 24, 2, seeds=[7, 5])
    assert both_t[1] == a_t[0] and both_l[1] == a_l[0]
    ranks = [e[2] for e in a_l[0]]
    assert min(ranks) >= 1 and max(ranks) > 1          # T = 0.9 on random weights: not always the top token
    for t, (ids, vals, rank) in zip(a_t[0], a_l[0]):
        assert ids[0] == t and vals[0] <= vals[1] and (rank == 1) == (ids[1] == t)
    eng = _engine(tw, dw, 3)
    eng.add_request(0, pa)
    eng.logprobs_top = 0
    with pytest.raises(RuntimeError):
        eng.step_overlapped()
    eng.step(sync=False)
    eng.step(sync=False)
    with pytest.raises(RuntimeError):
        eng.read_back()


def test_logprobs_through_the_v1_worker():
    """SamplingParams.logprobs on some requests of a step: ModelRunnerOutput.logprobs holds one
    row per emitted token of every request of the step, request by request; a prompt in pieces
    gets its first token's entry with the last piece; the values are the engine's."""
    from tests.test_gpu_worker import _worker
    from tests.test_host_logic import _NewRequestData, _sched
    tw, dw = _pair()
    prompts = {"a": [5, 9, 200, 31, 7], "b": list(range(40, 73)), "c": [int(t) for t in
               np.random.default_rng(1).integers(0, 256, 50)]}
    ar_t, ar_l = _run(_engine(tw, None, 0, max_batch=3), list(prompts.values()), 20, 2)
    want = {r: list(zip(ar_t[i], ar_l[i])) for i, r in enumerate(prompts)}
    w = _worker(tw, dw, 3, True)
    N = lambda r, n: _NewRequestData(r, prompts[r], sampling_params={"temperature": 0.0, "logprobs": n})
    got = {r: [] for r in prompts}

    def run(new=(), cached=(), pieces=None):
        so = _sched(new=list(new), cached=list(cached))
        so.num_scheduled_tokens.update(pieces or {})
        out = w.execute_model(so)
        lp = out.logprobs
        if lp is not None:
            assert len(lp.cu_num_generated_tokens) == len(out.req_ids) + 1
            assert lp.cu_num_generated_tokens[-1] == len(lp.logprobs) == len(lp.logprob_token_ids) \
                == len(lp.sampled_token_ranks) == sum(len(t) for t in out.sampled_token_ids)
        for i, (r, toks) in enumerate(zip(out.req_ids, out.sampled_token_ids)):
            for j, t in enumerate(toks):
                row = None
                if lp is not None:
                    n = lp.cu_num_generated_tokens[i] + j
                    row = (lp.logprob_token_ids[n], lp.logprobs[n], lp.sampled_token_ranks[n])
                got[r].append((t, row))
        return out

    out = run(new=[N("a", None)])
    assert out.logprobs is None                              # nobody asked
    out = run(new=[N("b", 2)], cached=["a"])                 # "b" asks: rows for the whole step
    assert out.logprobs is not None and len(out.logprobs.logprobs[0]) == 3
    run(new=[N("c", 1)], cached=["a", "b"], pieces={"c": 30})
    run(cached=["b", "a", "c"], pieces={"c": 20})            # last piece: first token and its entry
    while min(len(got[r]) for r in prompts) < 20:
        run(cached=["c", "a", "b"])
    for r in prompts:
        for n, ((t, row), (wt, wrow)) in enumerate(zip(got[r][:20], want[r])):
            assert t == wt, (r, n)
            if row is not None:
                assert row[0][:2] == wrow[0][:2] and row[1][:2] == wrow[1][:2] and row[2] == wrow[2], (r, n)
    assert all(row is not None for _, row in got["b"]) and got["a"][0][1] is None
    # a finished request stops asking
    so = _sched(cached=["a"], finished=["b", "c"])
    assert w.execute_model(so).logprobs is None and not w._logprobs_of


# ---------------------------------------------------------------------------- logit masks
def _masked_reference(tw, prompt, n_new, mask_row):
    """The autoregressive stream under a mask, computed step by step on the host from the
    engine's own logits (k = 0 engine, the mask applied here, greedy)."""
    eng = _engine(tw, None, 0)
    out = []
    eng.add_request(0, prompt)
    logits = eng.t_logits[0].clone()
    # an engine without a mask cannot be steered token by token, so the reference re-prefills
    seq = list(prompt)
    for _ in range(n_new):
        tok = int((logits + mask_row.to(logits.device)).argmax())
        out.append(tok)
        seq.append(tok)
        eng.reset()
        eng.add_request(0, seq)
        logits = eng.t_logits[0].clone()
    return out


@pytest.mark.parametrize("use_graph", [False, True])
def test_logit_masks_steer_the_target_and_stay_lossless(use_graph):
    """logit_bias / allowed_token_ids / banned tokens are added to the target's logits before
    argmax: a speculative engine emits the stream of masked autoregressive decoding (the draft
    sees no mask: its banned proposals are rejected), requests with and without masks share a
    batch, a mask follows its sequence to another slot, and the masked stream equals a
    reference that applies the mask on the host."""
    tw, dw = _pair()
    V = MODELS["llama"].vocab
    pa, pb, pc = [5, 9, 200, 31, 7], list(range(40, 60)), [3, 1, 4, 1, 5]
    allowed = list(range(0, V, 3))
    masks = [{"allowed_token_ids": allowed, "logit_bias": {6: 4.0, 9: -3.0}}, None,
             {"bad_token_ids": [0, 1, 2, 3, 4, 5, 6, 7], "logit_bias": {100: 2.5}}]
    row_a = torch.full((V,), float("-inf"))
    row_a[allowed] = 0.0
    row_a[6] += 4.0
    row_a[9] += -3.0
    want_a = _masked_reference(tw, pa, 14, row_a)
    assert all(t % 3 == 0 for t in want_a)
    kw = dict(use_graph=use_graph, max_batch=3)
    # autoregressive engine with the masks == the host reference; the unmasked request is untouched
    ar = _engine(tw, None, 0, **kw)
    outs = [[t] for t in ar.add_requests([pa, pb, pc], biases=masks)]
    while min(len(o) for o in outs) < 30:
        for o, new in zip(outs, ar.step()):
            o.extend(new)
    assert outs[0][:14] == want_a
    plain = _engine(tw, None, 0)
    ref_b = [plain.add_request(0, pb)]
    while len(ref_b) < 30:
        ref_b += plain.step()[0]
    assert outs[1][:30] == ref_b[:30]
    assert not set(outs[2][:30]) & set(range(8))
    # speculative engine: the same three streams; one at a time, in pieces, and after a move
    sp = _engine(tw, dw, 3, **kw)
    got = [[sp.add_request(0, pc, bias=masks[2])]]
    h = sp.prefill_begin(pa, bias=masks[0])
    assert sp.prefill_advance(h, 3) is None
    got[0] += sp.step()[0]
    got.append([sp.prefill_advance(h, 2)])
    got.append([sp.add_request(2, pb)])
    for _ in range(3):
        for o, new in zip(got, sp.step()):
            o.extend(new)
    sp.remove_request(0)                         # pb (no mask) moves from slot 2 to slot 0
    c_stream, got = got[0], [got[2], got[1]]
    while min(len(o) for o in got) < 30:
        for o, new in zip(got, sp.step()):
            o.extend(new)
    assert got[0][:30] == outs[1][:30] and got[1][:30] == outs[0][:30]
    assert c_stream == outs[2][:len(c_stream)]
    assert sp.metrics()["accepted"] >= 0
    with pytest.raises(ValueError):
        sp.add_request(2, pc, bias={"allowed_token_ids": [V + 5]})
    with pytest.raises(ValueError):
        sp.add_request(2, pc, bias={"allowed_token_ids": [3], "bad_token_ids": [3]})


def test_logit_masks_under_sampling_and_through_the_worker():
    """A sampling request with a whitelist only ever emits whitelisted tokens (accepted proposals,
    recovered and bonus tokens alike); through the V1 worker the masks arrive as sampling_params."""
    from tests.test_gpu_worker import _worker
    from tests.test_host_logic import _NewRequestData, _sched
    tw, dw = _pair()
    V = MODELS["llama"].vocab
    allowed = list(range(1, V, 2))
    eng = _engine(tw, dw, 3, max_batch=2, temperature=1.0, seed=4)
    outs = [[t] for t in eng.add_requests([[5, 9, 200], [7, 7, 7]], biases=[{"allowed_token_ids": allowed}, None])]
    for _ in range(40):
        for o, new in zip(outs, eng.step()):
            o.extend(new)
    assert all(t % 2 == 1 for t in outs[0]) and any(t % 2 == 0 for t in outs[1])
    w = _worker(tw, dw, 3, True)
    sp = {"temperature": 0.0, "allowed_token_ids": allowed, "logit_bias": {"11": 3.0},
          "bad_words_token_ids": [[13], [15]]}
    out = w.execute_model(_sched(new=[_NewRequestData("a", [5, 9, 200], sampling_params=sp),
                                      _NewRequestData("b", [5, 9, 200], sampling_params={"temperature": 0.0})]))
    toks = {r: list(t) for r, t in zip(out.req_ids, out.sampled_token_ids)}
    for _ in range(10):
        out = w.execute_model(_sched(cached=["a", "b"]))
        for r, t in zip(out.req_ids, out.sampled_token_ids):
            toks[r].extend(t)
    assert all(t % 2 == 1 and t not in (13, 15) for t in toks["a"]) and toks["a"] != toks["b"]


@pytest.mark.parametrize("use_graph", [False, True])
def test_logprobs_of_masked_requests_are_those_of_the_raw_logits(use_graph):
    """A request with a logit mask that asks for log-probabilities gets those of the target's RAW
    logits (upstream's sampler takes them before the logit processors run): the top entry is the
    raw arg-max, which the whitelist may forbid, and the emitted token's rank counts tokens the
    mask removed.  The speculative stream reports what the autoregressive stream reports."""
    tw, dw = _pair()
    V = MODELS["llama"].vocab
    allowed = list(range(0, V, 3))
    prompt = [5, 9, 200, 31, 7]

    def run(draft, k):
        eng = _engine(tw, draft, k, use_graph=use_graph)
        eng.logprobs_top = 1
        toks = [eng.add_request(0, prompt, bias={"allowed_token_ids": allowed})]
        raw = torch.log_softmax(eng.t_raw[0].double(), -1)
        ids, lps, rank = eng.last_logprobs[0][0]
        assert ids[0] == toks[0] and toks[0] % 3 == 0
        assert abs(lps[0] - raw[toks[0]].item()) < 1e-4
        assert ids[1] == int(raw.argmax()) and abs(lps[1] - raw.max().item()) < 1e-4
        assert rank == int((raw > raw[toks[0]]).sum()) + 1
        entries = [eng.last_logprobs[0][0]]
        while len(toks) < 24:
            new = eng.step()[0]
            toks += new
            entries += eng.last_logprobs[0]
        return toks[:24], entries[:24]

    ar_toks, ar_entries = run(None, 0)
    sp_toks, sp_entries = run(dw, 3)
    assert all(t % 3 == 0 for t in ar_toks) and sp_toks == ar_toks
    assert any(e[2] > 1 for e in ar_entries)         # the mask did overrule the raw arg-max somewhere
    for a, b in zip(ar_entries, sp_entries):
        assert a[0] == b[0] and a[2] == b[2] and a[1] == b[1]
