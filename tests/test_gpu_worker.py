"""The plugin surface (SpecDecodeWorker / Top1Proposer / MQAScorer /
RejectionSampler) on the GPU: same token streams as plain autoregressive
decoding, component path == graph path, sampler module == oracle."""
import numpy as np
import pytest
import torch

from oracle import sampling
from oracle.models import OracleConfig, random_state
from tests.test_gpu_model import MODELS, _bf, _engine, _generate, _product

pytestmark = pytest.mark.gpu


def _worker(tw, dw, k, use_graph, max_num_seqs=3, temperature=0.0, **kw):
    from vllm_omni_draft_amd.spec_decode_worker import SpecDecodeConfig, SpecDecodeWorker
    w = SpecDecodeWorker(SpecDecodeConfig(target=tw, draft=dw, num_speculative_tokens=k,
                                          max_num_seqs=max_num_seqs, max_model_len=200,
                                          split_tokens=128, use_graph=use_graph,
                                          temperature=temperature, **kw))
    w.init_device()
    w.load_model()
    assert w.determine_num_available_blocks()[0] > 0
    return w


@pytest.mark.parametrize("use_graph", [True, False])
def test_worker_streams_equal_autoregressive(use_graph):
    from vllm_omni_draft_amd.spec_decode_worker import ExecuteModelRequest, SequenceGroupMetadata
    tcfg = MODELS["llama"]
    dcfg = OracleConfig(**{**vars(MODELS["opt"]), "vocab": tcfg.vocab})
    tw = _product(tcfg, _bf(random_state(tcfg, seed=21)))
    dw = _product(dcfg, _bf(random_state(dcfg, seed=22)))
    prompts = {"r0": [5, 9, 200, 31, 7], "r1": list(range(40, 73)), "r2": [3, 3, 3]}
    want = {r: _generate(_engine(tw, None, 0), [p], 30)[0] for r, p in prompts.items()}
    w = _worker(tw, dw, 3, use_graph)
    got = {r: [] for r in prompts}
    out = w.execute_model(ExecuteModelRequest(
        [SequenceGroupMetadata(r, True, p) for r, p in list(prompts.items())[:2]]))
    for r, toks in zip(out.req_ids, out.sampled_token_ids):
        got[r].extend(toks)
    step = 0
    running = ["r0", "r1"]
    while min(len(got[r]) for r in prompts) < 30:
        groups = [SequenceGroupMetadata(r, False) for r in running]
        finished = []
        if step == 2:                       # a request joins mid-flight
            groups.append(SequenceGroupMetadata("r2", True, prompts["r2"]))
        if step == 6 and "r0" in running:   # and one leaves (its slot is reused)
            running.remove("r0")
            groups = [g for g in groups if g.request_id != "r0"]
            finished = ["r0"]
        out = w.execute_model(ExecuteModelRequest(groups, finished_requests_ids=finished))
        for r, toks in zip(out.req_ids, out.sampled_token_ids):
            assert 1 <= len(toks) <= 4
            got[r].extend(toks)
        if step == 2:
            running.append("r2")
        step += 1
        if "r0" not in running:
            got["r0"] = want["r0"]          # stopped early on purpose
    for r in prompts:
        assert got[r][:30] == want[r][:30], r
    m = w.metrics.maybe_collect(force=True)
    assert m.num_spec_tokens == 3 and m.emitted_tokens == m.accepted_tokens + m.draft_tokens // 3


def test_proposer_and_scorer_views():
    tcfg = MODELS["llama"]
    tw = _product(tcfg, _bf(random_state(tcfg, seed=21)))
    w = _worker(tw, tw, 3, use_graph=False, temperature=1.0)
    eng = w.engine
    eng.add_request(0, [1, 2, 3, 4])
    eng.add_request(1, [9, 8, 7])
    eng.prepare_step()
    props = w.proposer.get_spec_proposals(None, frozenset())
    assert props.proposal_token_ids.shape == (2, 3) and props.proposal_probs.shape == (2, 3, tcfg.vocab)
    assert props.proposal_lens.tolist() == [3, 3]
    scores = w.scorer.score_proposals(None, props)
    assert scores.probs.shape == (2, 4, tcfg.vocab) and scores.token_ids.shape == (2, 4)
    torch.cuda.synchronize()
    np.testing.assert_allclose(props.proposal_probs.sum(-1).cpu().numpy(), 1.0, rtol=1e-4)
    np.testing.assert_allclose(scores.probs.sum(-1).cpu().numpy(), 1.0, rtol=1e-4)
    # draft == target and same position => identical distributions row by row
    np.testing.assert_allclose(props.proposal_probs[:, 0].cpu().numpy(), scores.probs[:, 0].cpu().numpy(),
                               rtol=1e-5, atol=1e-8)
    w._verify_tokens(props, scores)
    emitted = eng.finish_step()
    assert all(1 <= len(t) <= 4 for t in emitted)


def test_rejection_sampler_module():
    from vllm_omni_draft_amd.spec_decode_worker import RejectionSampler
    rng = np.random.default_rng(3)
    B, k, V = 5, 4, 512
    p = sampling.softmax_rows(rng.standard_normal((B * (k + 1), V)).astype(np.float32) * 2).reshape(B, k + 1, V)
    q = sampling.softmax_rows(rng.standard_normal((B * k, V)).astype(np.float32) * 2).reshape(B, k, V)
    ids = np.array([[sampling.inverse_cdf(q[b, i], rng.random()) for i in range(k)] for b in range(B)])
    bonus = rng.integers(0, V, size=(B, 1))
    ua, ur = rng.random((B, k), dtype=np.float32), rng.random((B, k + 1), dtype=np.float32)
    rs = RejectionSampler(seed=1)
    cu = lambda a: torch.from_numpy(a).cuda()
    out = rs(cu(p), cu(bonus), cu(q), cu(ids), uniforms=(cu(ua), cu(ur)))
    ref, cnt = sampling.rejection_sample(p, bonus, q, ids, ua, ur)
    np.testing.assert_array_equal(out.cpu().numpy(), ref)
    rs.sync_counters()
    assert [rs.num_accepted_tokens, rs.num_emitted_tokens, rs.num_draft_tokens] == cnt.tolist()
    # generator-driven call: valid tokens, prefix property, seeded rows reproducible
    g = {2: torch.Generator(device="cuda").manual_seed(7)}
    a = rs(cu(p), cu(bonus), cu(q), cu(ids), seeded_seqs=g).cpu().numpy()
    g = {2: torch.Generator(device="cuda").manual_seed(7)}
    b = rs(cu(p), cu(bonus), cu(q), cu(ids), seeded_seqs=g).cpu().numpy()
    np.testing.assert_array_equal(a[2], b[2])
    for row, x in zip(a, ids):
        n = int((row >= 0).sum())
        assert 1 <= n <= k + 1 and (row[:n - 1] == x[:n - 1]).all() and (row[n:] == -1).all()


def test_typical_acceptance_sampler_module():
    from vllm_omni_draft_amd.spec_decode_worker import TypicalAcceptanceSampler
    g = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden",
                                           "typical_cases.npz"))
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    ts = TypicalAcceptanceSampler(0.09, 0.3)
    total = np.zeros(3, dtype=np.int64)
    for i in range(int(g["n_cases"])):
        if g[f"c{i}_proposal_lens"].size:      # the module's call shape has no ragged lengths
            continue
        out = ts(cu(g[f"c{i}_target_probs"]), cu(g[f"c{i}_bonus_ids"]), None, cu(g[f"c{i}_draft_ids"]))
        np.testing.assert_array_equal(out.cpu().numpy(), g[f"c{i}_out_ids"], err_msg=f"case {i}")
        total += g[f"c{i}_counters"]
    ts.sync_counters()
    assert [ts.num_accepted_tokens, ts.num_emitted_tokens, ts.num_draft_tokens] == total.tolist()


@pytest.mark.parametrize("option", ["ngram", "disable_by_batch_size", "typical_never", "fp8", "draft_tree"])
def test_worker_options_keep_the_greedy_stream(option):
    """Options that keep upstream's names, through the worker: prompt lookup without a draft
    model, speculation switched off above a batch size, a typical-acceptance threshold nothing
    passes (every step emits the target's most likely token), fp8 weights, a tree of three
    branches grown by a (rotary) draft model: the stream stays the target's own greedy stream."""
    from vllm_omni_draft_amd.spec_decode_worker import (ExecuteModelRequest, SequenceGroupMetadata,
                                                        SpecDecodeConfig, SpecDecodeWorker)
    tcfg = MODELS["llama"]
    dcfg = OracleConfig(**{**vars(MODELS["opt"]), "vocab": tcfg.vocab})
    if option == "draft_tree":
        dcfg = OracleConfig(**{**vars(tcfg), "n_layers": 1})
    fp8 = option == "fp8"
    tw = _product(tcfg, _bf(random_state(tcfg, seed=21)), fp8=fp8)
    dw = _product(dcfg, _bf(random_state(dcfg, seed=22)), fp8=fp8)
    prompts = {"a": [5, 9, 200, 31, 7, 5, 9, 200], "b": list(range(40, 60))}
    want = {r: _generate(_engine(tw, None, 0), [p], 20)[0] for r, p in prompts.items()}
    kw = dict(target=tw, draft=dw, num_speculative_tokens=3, max_num_seqs=2, max_model_len=200,
              split_tokens=128)
    if option == "ngram":
        kw.update(draft=None, ngram_prompt_lookup_max=3, ngram_prompt_lookup_min=1)
    elif option == "disable_by_batch_size":
        kw.update(speculative_disable_by_batch_size=1)
    elif option == "draft_tree":
        kw.update(speculative_draft_tree_branches=3)
    elif option == "typical_never":
        kw.update(temperature=1.0, spec_decoding_acceptance_method="typical_acceptance_sampler",
                  typical_acceptance_sampler_posterior_threshold=2.0,
                  typical_acceptance_sampler_posterior_alpha=1e9)
    w = SpecDecodeWorker(SpecDecodeConfig(**kw))
    w.init_device()
    w.load_model()
    if option == "draft_tree":
        assert w.engine.tree and w.engine.tree_nodes == 9
    got = {r: [] for r in prompts}
    out = w.execute_model(ExecuteModelRequest(
        [SequenceGroupMetadata(r, True, p) for r, p in prompts.items()]))
    for r, toks in zip(out.req_ids, out.sampled_token_ids):
        got[r].extend(toks)
    while min(len(t) for t in got.values()) < 20:
        out = w.execute_model(ExecuteModelRequest([SequenceGroupMetadata(r, False) for r in prompts]))
        for r, toks in zip(out.req_ids, out.sampled_token_ids):
            if option in ("disable_by_batch_size", "typical_never"):
                assert len(toks) == 1
            got[r].extend(toks)
    if option == "typical_never":
        # the first token of each prompt is sampled; what follows is the greedy continuation
        for r, p in prompts.items():
            cont = _generate(_engine(tw, None, 0), [p + got[r][:1]], 19)[0]
            assert got[r][1:20] == cont
    else:
        for r in prompts:
            assert got[r][:20] == want[r], r


def test_worker_loads_and_shards_from_state_dicts_with_a_group_of_one():
    """target and draft given as (config, state dict): load_model builds (and, under tensor
    parallelism, shards) both itself.  With an RCCL group of one rank, vocabulary-parallel heads
    and the step graph with the collectives inside, the stream is the plain worker's."""
    from vllm_omni_draft_amd.spec_decode_worker import (ExecuteModelRequest, SequenceGroupMetadata,
                                                        SpecDecodeConfig, SpecDecodeWorker)
    tcfg = MODELS["llama"]
    dcfg = OracleConfig(**{**vars(MODELS["llama"]), "n_layers": 1})
    tstate, dstate = _bf(random_state(tcfg, seed=21)), _bf(random_state(dcfg, seed=23))
    tw, dw = _product(tcfg, tstate), _product(dcfg, dstate)
    prompt = [5, 9, 200, 31, 7]

    def run(w):
        out = w.execute_model(ExecuteModelRequest([SequenceGroupMetadata("r", True, prompt)])).sampled_token_ids[0]
        while len(out) < 24:
            out += w.execute_model(ExecuteModelRequest([SequenceGroupMetadata("r", False)])).sampled_token_ids[0]
        return out[:24]

    want = run(_worker(tw, dw, 3, True, max_num_seqs=1))
    # tensor_parallel_size 1 through the tuple path
    w1 = SpecDecodeWorker(SpecDecodeConfig(target=(tw.cfg, tstate), draft=(dw.cfg, dstate),
                                           num_speculative_tokens=3, max_num_seqs=1, max_model_len=200,
                                           split_tokens=128))
    w1.init_device()
    w1.load_model()
    assert run(w1) == want
    # the tensor-parallel engine with a group of one rank, both heads split, graph on
    from vllm_omni_draft_amd import parallel, weights
    from vllm_omni_draft_amd.engine import EngineConfig, SpecEngine
    group = parallel.TensorParallelGroup(0, 1)
    lt, st = weights.shard_state_dict(tw.cfg, tstate, 1, 0, vocab_parallel=True)
    ld, sd = weights.shard_state_dict(dw.cfg, dstate, 1, 0, vocab_parallel=True)
    eng = SpecEngine(weights.from_state_dict(lt, st, "cuda"), weights.from_state_dict(ld, sd, "cuda"),
                     EngineConfig(k=3, max_batch=1, max_seq_len=200, split_tokens=128),
                     tp_group=group, draft_tp=True)
    assert eng.vp and eng.d_vp
    w2 = SpecDecodeWorker(vllm_config=None, engine=eng)
    assert run(w2) == want and eng._graphs
    del w2, eng
    group.close()


def test_fp8_draft_under_a_bf16_target_keeps_the_target_stream():
    """draft_quantization="fp8": the draft's matrices are quantised at load (half the bytes per
    draft pass), the target stays bf16, so the greedy stream is still the bf16 target's own
    autoregressive stream; a draft that is the target's fp8 copy still gets proposals accepted.
    Options that do not match the weights they are given for are refused."""
    from vllm_omni_draft_amd.spec_decode_worker import (ExecuteModelRequest, SequenceGroupMetadata,
                                                        SpecDecodeConfig, SpecDecodeWorker)
    tcfg = MODELS["llama"]
    tstate = _bf(random_state(tcfg, seed=21))
    tw = _product(tcfg, tstate)
    prompt = [5, 9, 200, 31, 7]
    want = _generate(_engine(tw, None, 0), [prompt], 30)[0]
    w = SpecDecodeWorker(SpecDecodeConfig(target=tw, draft=(tw.cfg, tstate), draft_quantization="fp8",
                                          num_speculative_tokens=3, max_num_seqs=1, max_model_len=200,
                                          split_tokens=128))
    w.init_device()
    w.load_model()
    assert w.engine.draft.weights.fp8 and not w.engine.target.weights.fp8
    assert w.engine.draft.weights.layers[0].qkv_w.scale is not None
    out = w.execute_model(ExecuteModelRequest([SequenceGroupMetadata("r", True, prompt)])).sampled_token_ids[0]
    while len(out) < 30:
        out += w.execute_model(ExecuteModelRequest([SequenceGroupMetadata("r", False)])).sampled_token_ids[0]
    assert out[:30] == want
    m = w.metrics.maybe_collect(force=True)
    assert m.accepted_tokens > m.draft_tokens // 2      # an fp8 copy of the target mostly agrees with it
    # the same draft with an fp8 head over half of the vocabulary: still the target's stream
    w2 = SpecDecodeWorker(SpecDecodeConfig(target=tw, draft=(tw.cfg, tstate), draft_quantization="fp8",
                                           draft_vocab_subset=list(range(0, tcfg.vocab, 2)),
                                           num_speculative_tokens=3, max_num_seqs=1, max_model_len=200,
                                           split_tokens=128))
    w2.init_device()
    w2.load_model()
    assert w2.engine.draft.cfg.lm_vocab == tcfg.vocab // 2 and w2.engine.d_map is not None
    out2 = w2.execute_model(ExecuteModelRequest([SequenceGroupMetadata("r", True, prompt)])).sampled_token_ids[0]
    while len(out2) < 30:
        out2 += w2.execute_model(ExecuteModelRequest([SequenceGroupMetadata("r", False)])).sampled_token_ids[0]
    assert out2[:30] == want
    for bad in (dict(draft=tw, draft_quantization="fp8"),           # bf16 weights object
                dict(draft=tw, draft_vocab_subset=list(range(16))),  # a whole-vocabulary weights object
                dict(draft=(tw.cfg, tstate), draft_quantization="int4"),
                dict(draft=None, draft_quantization="fp8"),
                dict(draft=(tw.cfg, tstate), quantization="fp8")):  # the target object is bf16
        wb = SpecDecodeWorker(SpecDecodeConfig(target=tw, num_speculative_tokens=3, max_num_seqs=1,
                                               max_model_len=200, split_tokens=128, **bad))
        wb.init_device()
        with pytest.raises(ValueError):
            wb.load_model()


def test_worker_serves_requests_at_their_own_temperature_and_seed():
    """SamplingParams.temperature and .seed per request through the worker: on a sampling worker
    a temperature-0 request gets the greedy stream and a seeded request the same tokens whether
    it runs alone or beside others; a greedy worker refuses a request that wants to sample."""
    from vllm_omni_draft_amd.spec_decode_worker import (ExecuteModelRequest, SamplingParams,
                                                        SequenceGroupMetadata)
    tcfg = MODELS["llama"]
    dcfg = OracleConfig(**{**vars(MODELS["opt"]), "vocab": tcfg.vocab})
    tw = _product(tcfg, _bf(random_state(tcfg, seed=21)))
    dw = _product(dcfg, _bf(random_state(dcfg, seed=22)))
    pa, pb = [5, 9, 200, 31, 7], list(range(40, 60))
    greedy = _generate(_engine(tw, None, 0), [pa], 24)[0]

    def run(worker, reqs):
        out = worker.execute_model(ExecuteModelRequest(
            [SequenceGroupMetadata(r, True, p, SamplingParams(**sp)) for r, p, sp in reqs]))
        got = {r: list(t) for r, t in zip(out.req_ids, out.sampled_token_ids)}
        while min(len(t) for t in got.values()) < 24:
            out = worker.execute_model(ExecuteModelRequest(
                [SequenceGroupMetadata(r, False) for r, _, _ in reqs]))
            for r, t in zip(out.req_ids, out.sampled_token_ids):
                got[r] += t
        return {r: t[:24] for r, t in got.items()}

    alone = run(_worker(tw, dw, 3, True, max_num_seqs=1, temperature=0.8),
                [("b", pb, dict(seed=7, temperature=1.2))])["b"]
    both = run(_worker(tw, dw, 3, True, max_num_seqs=2, temperature=0.8),
               [("a", pa, dict(temperature=0.0)), ("b", pb, dict(seed=7, temperature=1.2))])
    assert both["a"] == greedy and both["b"] == alone
    with pytest.raises(ValueError):
        run(_worker(tw, dw, 3, True, max_num_seqs=1), [("a", pa, dict(temperature=0.8))])


def _llama_pair():
    tcfg = MODELS["llama"]
    dcfg = OracleConfig(**{**vars(MODELS["opt"]), "vocab": tcfg.vocab})
    return (_product(tcfg, _bf(random_state(tcfg, seed=21))),
            _product(dcfg, _bf(random_state(dcfg, seed=22))))


@pytest.mark.parametrize("use_graph", [True, False])
def test_v1_scheduler_output_path_gives_the_autoregressive_streams(use_graph):
    """The worker driven the way the reference's workers are driven
    (execute_model(scheduler_output, delay_capture), /root/reference/src/qwen_image_gen/worker.py:47):
    new requests, cached requests, a request that joins and one that is finished mid-flight."""
    from tests.test_host_logic import _NewRequestData, _sched
    tw, dw = _llama_pair()
    prompts = {"r0": [5, 9, 200, 31, 7], "r1": list(range(40, 73)), "r2": [3, 3, 3]}
    want = {r: _generate(_engine(tw, None, 0), [p], 30)[0] for r, p in prompts.items()}
    w = _worker(tw, dw, 3, use_graph)
    got = {r: [] for r in prompts}

    def take(out):
        assert out.logprobs is None and len(out.pooler_output) == len(out.req_ids)
        for r, toks in zip(out.req_ids, out.sampled_token_ids):
            assert 1 <= len(toks) <= 4
            got[r].extend(toks)

    N = lambda r: _NewRequestData(r, prompts[r], sampling_params={"temperature": 0.0, "max_tokens": 30})
    take(w.execute_model(_sched(new=[N("r0"), N("r1")]), delay_capture=False))
    running, step, finished = ["r0", "r1"], 0, []
    while min(len(got[r]) for r in ("r1", "r2")) < 30:
        new = [N("r2")] if step == 2 else []
        if step == 6:
            running.remove("r0")
            finished = ["r0"]
        take(w.execute_model(_sched(new=new, cached=running, finished=finished,
                                    old_layout=bool(step & 1))))
        finished = []
        if step == 2:
            running.append("r2")
        step += 1
    assert len(got["r0"]) >= 7
    for r in prompts:
        n = min(30, len(got[r]))
        assert got[r][:n] == want[r][:n], r


def test_v1_chunked_prefill_through_the_worker():
    """A V1 scheduler with chunked prefill: a long prompt arrives as num_scheduled_tokens pieces
    over several steps (first as a new request, then among the cached ones) while another request
    keeps decoding; a piece emits no token, the last one the first token; a prompt can be dropped
    half-way.  Streams equal the autoregressive ones."""
    from tests.test_host_logic import _NewRequestData, _sched
    tw, dw = _llama_pair()
    prompts = {"r0": [5, 9, 200, 31, 7], "long": [int(t) for t in np.random.default_rng(1).integers(0, 256, 90)],
               "gone": list(range(60, 110))}
    want = {r: _generate(_engine(tw, None, 0), [p], 24)[0] for r, p in prompts.items()}
    w = _worker(tw, dw, 3, True)
    got = {r: [] for r in prompts}
    N = lambda r: _NewRequestData(r, prompts[r], sampling_params={"temperature": 0.0})

    def run(new=(), cached=(), pieces=None, finished=(), resumed=()):
        so = _sched(new=list(new), cached=list(cached), finished=finished, resumed=resumed)
        so.num_scheduled_tokens.update(pieces or {})
        out = w.execute_model(so)
        for r, toks in zip(out.req_ids, out.sampled_token_ids):
            got[r].extend(toks)
        return dict(zip(out.req_ids, out.sampled_token_ids))

    run(new=[N("r0")])
    o = run(new=[N("long")], cached=["r0"], pieces={"long": 40})          # first piece
    assert o["long"] == [] and len(o["r0"]) >= 1
    o = run(new=[N("gone")], cached=["r0", "long"], pieces={"long": 30, "gone": 20})
    assert o["long"] == [] and o["gone"] == []
    with pytest.raises(ValueError):
        run(cached=["r0", "long"], pieces={"long": 21})                    # more than what remains
    o = run(cached=["r0", "long"], pieces={"long": 20}, finished=["gone"])  # last piece; "gone" dropped
    assert len(o["long"]) == 1 and w.engine.active == 2 and not w._prefill_of
    while min(len(got[r]) for r in ("r0", "long")) < 24:
        run(cached=["long", "r0"])
    for r in ("r0", "long"):
        assert got[r][:24] == want[r], r
    assert got["gone"] == []


class _StubProposer:
    """A proposer made outside the worker: it knows the target's true stream and proposes a
    continuation of `lens[r]` tokens for request r, with an error injected at position
    `wrong_at[r]` (None: none)."""

    def __init__(self, want, prompts, k):
        self.want, self.prompts, self.k = want, prompts, k
        self.done = {}          # request -> tokens emitted so far
        self.order = []
        self.plan = {}          # request -> (len, wrong_at)
        self.calls = 0

    def get_spec_proposals(self, execute_model_req, seq_ids_with_bonus_token_in_last_step):
        from vllm_omni_draft_amd.spec_decode_worker import SpeculativeProposals
        self.calls += 1
        B, k = len(self.order), self.k
        ids = torch.zeros(B, k, dtype=torch.int64)
        lens = torch.zeros(B, dtype=torch.int32)
        for b, r in enumerate(self.order):
            n, wrong = self.plan[r]
            cont = list(self.want[r][self.done[r]:self.done[r] + n])
            if wrong is not None and wrong < len(cont):
                cont[wrong] = (cont[wrong] + 1) % 256
            lens[b] = len(cont)
            ids[b, :len(cont)] = torch.tensor(cont, dtype=torch.int64)
        return SpeculativeProposals(ids.cuda(), None, lens.cuda())


def test_a_foreign_proposer_plugs_into_the_scorer():
    """MQAScorer.score_proposals verifies proposals it did not make: full, ragged, empty and
    partly wrong continuations; the emitted streams are the target's autoregressive ones and
    every step emits exactly (accepted prefix + 1) tokens."""
    from vllm_omni_draft_amd.spec_decode_worker import (ExecuteModelRequest, SequenceGroupMetadata,
                                                        SpecDecodeConfig, SpecDecodeWorker)
    tw, _ = _llama_pair()
    k = 4
    prompts = {"a": [5, 9, 200, 31, 7], "b": list(range(40, 73)), "c": [3, 3, 3]}
    want = {r: _generate(_engine(tw, None, 0), [p], 40)[0] for r, p in prompts.items()}
    stub = _StubProposer(want, prompts, k)
    w = SpecDecodeWorker(SpecDecodeConfig(target=tw, draft=None, num_speculative_tokens=k,
                                          external_proposals=True, max_num_seqs=3, max_model_len=200,
                                          split_tokens=128), proposer=stub)
    w.init_device()
    w.load_model()
    out = w.execute_model(ExecuteModelRequest([SequenceGroupMetadata(r, True, p)
                                               for r, p in prompts.items()]))
    got = {r: list(t) for r, t in zip(out.req_ids, out.sampled_token_ids)}
    stub.order = list(prompts)
    plans = [{"a": (4, None), "b": (2, None), "c": (0, None)},      # full, ragged, empty
             {"a": (4, 1), "b": (4, 0), "c": (3, 2)},               # wrong at 1, at 0, at 2
             {"a": (1, None), "b": (4, 3), "c": (4, None)}]
    for step in range(9):
        stub.plan = plans[step % 3]
        stub.done = {r: len(got[r]) for r in prompts}
        out = w.execute_model(ExecuteModelRequest([SequenceGroupMetadata(r, False) for r in prompts]))
        for r, toks in zip(out.req_ids, out.sampled_token_ids):
            n, wrong = stub.plan[r]
            assert len(toks) == (n if wrong is None else min(wrong, n)) + 1, (step, r, toks)
            got[r].extend(toks)
    assert stub.calls == 9
    for r in prompts:
        assert got[r] == want[r][:len(got[r])], r


@pytest.mark.parametrize("use_graph", [True, False])
def test_scheduler_supplied_spec_tokens_are_verified(use_graph):
    """A V1 scheduler output's scheduled_spec_decode_tokens (proposals made by the engine above
    the worker) are what an external_proposals worker verifies in that step."""
    from tests.test_host_logic import _NewRequestData, _sched
    from vllm_omni_draft_amd.spec_decode_worker import SpecDecodeConfig, SpecDecodeWorker
    tw, _ = _llama_pair()
    k = 3
    prompts = {"a": [5, 9, 200, 31, 7], "b": list(range(40, 73))}
    want = {r: _generate(_engine(tw, None, 0), [p], 40)[0] for r, p in prompts.items()}
    w = SpecDecodeWorker(SpecDecodeConfig(target=tw, draft=None, num_speculative_tokens=k,
                                          external_proposals=True, max_num_seqs=2, max_model_len=200,
                                          split_tokens=128, use_graph=use_graph))
    w.init_device()
    w.load_model()
    out = w.execute_model(_sched(new=[_NewRequestData(r, p) for r, p in prompts.items()]))
    got = {r: list(t) for r, t in zip(out.req_ids, out.sampled_token_ids)}
    for step in range(8):
        spec = {}
        if step % 4 != 3:                                   # every fourth step carries no proposals
            spec["a"] = want["a"][len(got["a"]):len(got["a"]) + k]
        if step % 2 == 0:
            spec["b"] = [(t + 1) % 256 for t in want["b"][len(got["b"]):len(got["b"]) + 2]]
        out = w.execute_model(_sched(cached=list(prompts), spec=spec))
        for r, toks in zip(out.req_ids, out.sampled_token_ids):
            want_n = (k + 1) if (r == "a" and "a" in spec) else 1
            assert len(toks) == want_n, (step, r, toks)
            got[r].extend(toks)
    for r in prompts:
        assert got[r] == want[r][:len(got[r])], r


def test_metrics_are_copied_on_a_side_stream():
    """AsyncMetricsCollector on the GPU: the interval starts a copy of the counters behind the
    steps enqueued so far, a later step returns them; they equal a synchronous read."""
    import time
    from vllm_omni_draft_amd.spec_decode_worker import ExecuteModelRequest, SequenceGroupMetadata
    tw, dw = _llama_pair()
    w = _worker(tw, dw, 3, True)
    w.metrics.interval = 0.0
    out = w.execute_model(ExecuteModelRequest([SequenceGroupMetadata("a", True, [5, 9, 200, 31, 7])]))
    assert out.spec_decode_metrics is None            # the copy was only started
    seen = None
    for _ in range(50):
        out = w.execute_model(ExecuteModelRequest([SequenceGroupMetadata("a", False)]))
        if out.spec_decode_metrics is not None and out.spec_decode_metrics.draft_tokens > 0:
            seen = out.spec_decode_metrics           # (the first copy predates the first step)
            break
        time.sleep(0.001)
    assert seen is not None and seen.num_spec_tokens == 3
    now = w.metrics.maybe_collect(force=True)
    assert 0 < seen.draft_tokens <= now.draft_tokens and seen.emitted_tokens <= now.emitted_tokens
    assert seen.draft_tokens % 3 == 0


def test_v1_preempted_requests_are_rebuilt_when_they_return():
    """A V1 scheduler preempts a request by not scheduling it and later names it again among the
    cached requests: the worker frees its slot, keeps its tokens, and prefills prompt + emitted
    tokens anew when it returns (whole, or in the pieces the scheduler grants).  Greedy streams are
    the autoregressive ones, interruptions or not (bit for bit on the wide kernel family, where
    decode rows and rebuilt rows come from the same kernels)."""
    from tests.test_host_logic import _NewRequestData, _sched
    tw, dw = _llama_pair()
    prompts = {"a": [5, 9, 200, 31, 7], "b": list(range(40, 73)), "c": [3, 3, 3]}
    want = {r: _generate(_engine(tw, None, 0, rows="wide"), [p], 40)[0] for r, p in prompts.items()}
    w = _worker(tw, dw, 3, True, rows="wide")
    got = {r: [] for r in prompts}
    N = lambda r: _NewRequestData(r, prompts[r], sampling_params={"temperature": 0.0})

    def run(new=(), cached=(), pieces=None, finished=(), resumed=()):
        so = _sched(new=list(new), cached=list(cached), finished=finished, resumed=resumed)
        so.num_scheduled_tokens.update(pieces or {})
        out = w.execute_model(so)
        for r, toks in zip(out.req_ids, out.sampled_token_ids):
            got[r].extend(toks)
        return dict(zip(out.req_ids, out.sampled_token_ids))

    run(new=[N("a"), N("b"), N("c")])
    run(cached=["a", "b", "c"])
    run(cached=["a", "c"])                               # "b" is preempted
    assert w.engine.active == 2 and "b" in w._parked
    run(cached=["c"])                                    # and "a"
    assert w.engine.active == 1 and w._parked == {"a", "b"}
    n_b = len(prompts["b"]) + len(got["b"])
    o = run(cached=["c", "b"], pieces={"b": n_b}, resumed={"b"})        # "b" returns whole
    assert len(o["b"]) == 1 and w.engine.active == 2
    n_a = len(prompts["a"]) + len(got["a"])
    o = run(cached=["c", "b", "a"], pieces={"a": n_a - 3}, resumed={"a"})   # "a" returns in two pieces
    assert o["a"] == []
    o = run(cached=["c", "b", "a"], pieces={"a": 3})
    assert len(o["a"]) == 1 and w.engine.active == 3 and not w._parked
    while min(len(got[r]) for r in prompts) < 40:
        run(cached=["a", "b", "c"])
    for r in prompts:
        assert got[r][:40] == want[r], r
    run(cached=["a"], finished=["b"])                    # "c" parked, "b" finished
    run(cached=["a"], finished=["c"])                    # finished while parked
    assert w.engine.active == 1 and set(w._tokens) == {"a"}


@pytest.mark.parametrize("rows", ["wide", "narrow"])
def test_v1_requests_restarted_without_a_step_in_between(rows):
    """A scheduler that preempts a request and resumes it in its very next step (nothing else was
    running, so no step went by without it) only sets resumed_from_preemption with zero computed
    tokens: the worker takes the sequence out of its slot itself and rebuilds it, whole or in
    pieces.  The same flag on a request whose prompt was still arriving in pieces starts its
    prefill over.  Streams stay the autoregressive ones, bit for bit."""
    from tests.test_host_logic import _NewRequestData, _sched
    tw, dw = _llama_pair()
    prompts = {"a": [5, 9, 200, 31, 7, 11, 2], "b": list(range(40, 73))}
    want = {r: _generate(_engine(tw, None, 0, rows=rows), [p], 30)[0] for r, p in prompts.items()}
    w = _worker(tw, dw, 3, True, rows=rows)
    got = {r: [] for r in prompts}
    N = lambda r: _NewRequestData(r, prompts[r], sampling_params={"temperature": 0.0})

    def run(new=(), cached=(), pieces=None, finished=(), resumed=()):
        so = _sched(new=list(new), cached=list(cached), finished=finished, resumed=resumed)
        so.num_scheduled_tokens.update(pieces or {})
        out = w.execute_model(so)
        for r, toks in zip(out.req_ids, out.sampled_token_ids):
            got[r].extend(toks)
        return dict(zip(out.req_ids, out.sampled_token_ids))

    run(new=[N("a")])
    run(cached=["a"])
    run(cached=["a"])
    n_a = len(prompts["a"]) + len(got["a"])
    o = run(cached=["a"], pieces={"a": n_a}, resumed={"a"})             # restarted whole, never left out
    assert len(o["a"]) == 1 and w.engine.active == 1 and not w._parked
    run(cached=["a"])
    n_a = len(prompts["a"]) + len(got["a"])
    o = run(cached=["a"], pieces={"a": n_a - 4}, resumed={"a"})         # restarted in two pieces
    assert o["a"] == [] and w.engine.active == 0 and len(w._prefill_of) == 1
    o = run(cached=["a"], pieces={"a": 4})
    assert len(o["a"]) == 1 and w.engine.active == 1
    # "b" arrives in pieces and is restarted half way: its prefill starts over, in other pieces
    o = run(new=[N("b")], cached=["a"], pieces={"b": 10})
    assert o["b"] == [] and len(w._prefill_of) == 1
    o = run(cached=["a", "b"], pieces={"b": 20}, resumed={"b"})
    assert o["b"] == [] and len(w._prefill_of) == 1 and w._prefill_left["b"] == len(prompts["b"]) - 20
    o = run(cached=["a", "b"], pieces={"b": len(prompts["b"]) - 20})
    assert len(o["b"]) == 1 and w.engine.active == 2 and not w._prefill_of
    while min(len(got[r]) for r in prompts) < 30:
        run(cached=["a", "b"])
    for r in prompts:
        assert got[r][:30] == want[r], r


@pytest.mark.parametrize("kind", ["draft_tree", "lookup_tree"])
def test_tree_workers_step_as_a_whole_without_the_graph(kind):
    """A tree engine has no [B, k] proposal to hand to the scorer, so the worker runs its step as
    one unit also when graphs are off (the piecewise propose / score / accept path serves chains).
    Seen from outside: trees are really verified (with draft == target every step accepts k
    tokens; prompt lookup accepts some), and the log-probability report names the emitted tokens
    (its rows follow the accepted path of the tree step that ran)."""
    from vllm_omni_draft_amd.spec_decode_worker import (ExecuteModelRequest, SamplingParams,
                                                        SequenceGroupMetadata, SpecDecodeConfig,
                                                        SpecDecodeWorker)
    tcfg = MODELS["llama"]
    tw = _product(tcfg, _bf(random_state(tcfg, seed=21)))
    prompts = {"a": [5, 9, 200, 31, 7, 5, 9, 200, 31], "b": list(range(40, 52)) * 2}
    want = {r: _generate(_engine(tw, None, 0), [p], 24)[0] for r, p in prompts.items()}
    kw = dict(target=tw, draft=tw, num_speculative_tokens=3, max_num_seqs=2, max_model_len=200,
              split_tokens=128, use_graph=False)
    if kind == "draft_tree":
        kw.update(speculative_draft_tree_branches=3)
    else:
        kw.update(draft=None, ngram_prompt_lookup_max=3, ngram_prompt_lookup_min=1, speculative_tree_candidates=3)
    w = SpecDecodeWorker(SpecDecodeConfig(**kw))
    w.init_device()
    w.load_model()
    assert w.engine.tree
    sp = SamplingParams(temperature=0.0, logprobs=1)
    got = {r: [] for r in prompts}
    out = w.execute_model(ExecuteModelRequest([SequenceGroupMetadata(r, True, p, sp) for r, p in prompts.items()]))
    steps = 0
    while True:
        lp = out.logprobs
        for i, (r, toks) in enumerate(zip(out.req_ids, out.sampled_token_ids)):
            for j, t in enumerate(toks):
                n = lp.cu_num_generated_tokens[i] + j
                assert lp.logprob_token_ids[n][0] == t and lp.sampled_token_ranks[n] == 1, (r, j)
            got[r].extend(toks)
        if min(len(t) for t in got.values()) >= 24:
            break
        out = w.execute_model(ExecuteModelRequest([SequenceGroupMetadata(r, False) for r in prompts]))
        steps += 1
    for r in prompts:
        assert got[r][:24] == want[r], r
    m = w.engine.metrics()
    if kind == "draft_tree":
        assert m["accepted"] == 3 * 2 * steps, m          # the first branch is the target's own continuation
    else:
        assert m["accepted"] > 0, m
