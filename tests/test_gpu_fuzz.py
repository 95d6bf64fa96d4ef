"""Randomised schedules on the GPU: requests join and leave a running batch at random steps.
Whatever the schedule, every request's greedy stream must equal the stream the same request
gets alone from an autoregressive engine of the same kernel family (rows are independent of
their batch neighbours bit for bit; speculation is lossless)."""
import random

import pytest
import torch

from oracle.models import OracleConfig, random_state

pytestmark = pytest.mark.gpu

TARGET = OracleConfig(arch="llama", n_layers=2, hidden=256, n_heads=8, n_kv_heads=2, head_dim=64,
                      ffn=512, vocab=512, max_pos=512)
DRAFT = OracleConfig(arch="llama", n_layers=1, hidden=128, n_heads=2, n_kv_heads=1, head_dim=64,
                     ffn=256, vocab=512, max_pos=512, tie_embeddings=True)


def _weights(oc, seed):
    from vllm_omni_draft_amd import config, weights
    cfg = config.create_model_config(
        None, arch=oc.arch, n_layers=oc.n_layers, hidden=oc.hidden, n_heads=oc.n_heads,
        n_kv_heads=oc.n_kv_heads, head_dim=oc.head_dim, ffn=oc.ffn, vocab=oc.vocab,
        norm_eps=oc.norm_eps, rope_theta=oc.rope_theta, max_pos=oc.max_pos,
        tie_embeddings=oc.tie_embeddings)
    state = {k: v.to(torch.bfloat16) for k, v in random_state(oc, seed=seed).items()}
    return weights.from_state_dict(cfg, state, "cuda")


@pytest.mark.parametrize("rows,max_batch,k,use_graph,seed", [
    ("narrow", 2, 5, True, 0), ("narrow", 2, 3, False, 1), ("wide", 6, 3, True, 2),
    ("wide", 6, 5, False, 3), ("wide", 20, 2, True, 4)])
def test_random_join_leave_schedule(rows, max_batch, k, use_graph, seed):
    from vllm_omni_draft_amd.engine import EngineConfig, SpecEngine
    rnd = random.Random(seed)
    tw, dw = _weights(TARGET, 50), _weights(DRAFT, 51)
    n_req, n_new = 3 * max_batch if max_batch <= 6 else 30, 28
    prompts = [[rnd.randrange(TARGET.vocab) for _ in range(rnd.randint(1, 70))] for _ in range(n_req)]

    def cfg(kk, mb):
        return EngineConfig(k=kk, max_batch=mb, max_seq_len=140, split_tokens=128, use_graph=use_graph,
                            rows=rows)

    # every request alone, autoregressive, same kernel family
    solo = SpecEngine(tw, None, cfg(0, 1))
    want = []
    for p in prompts:
        solo.reset()
        out = [solo.add_request(0, p)]
        while len(out) < n_new:
            out.extend(solo.step()[0])
        want.append(out[:n_new])
    del solo

    eng = SpecEngine(tw, dw, cfg(k, max_batch))
    waiting = list(range(n_req))
    running = []           # request index per slot
    got = {i: [] for i in range(n_req)}
    steps = 0
    while waiting or running:
        # admit a random number of waiting requests into the free slots
        while waiting and len(running) < max_batch and (not running or rnd.random() < 0.6):
            i = waiting.pop(rnd.randrange(len(waiting)))
            got[i].append(eng.add_request(len(running), prompts[i]))
            running.append(i)
        for slot, new in enumerate(eng.step()):
            got[running[slot]].extend(new)
        steps += 1
        # finished requests leave (highest slot first: the last sequence moves into the hole)
        for slot in sorted((s for s, i in enumerate(running) if len(got[i]) >= n_new), reverse=True):
            eng.remove_request(slot)
            last = len(running) - 1
            if slot != last:
                running[slot] = running[last]
            running.pop()
        assert steps < 2000
    for i in range(n_req):
        assert got[i][:n_new] == want[i], i
    m = eng.metrics()
    assert m["draft"] > 0 and m["emitted"] >= m["accepted"]


def test_scheduler_generates_what_each_request_gets_alone():
    """SpecDecodeScheduler over a real worker: 7 requests through 3 slots, different budgets and
    an end-of-sequence token; every output is the solo autoregressive stream cut at its stop."""
    from vllm_omni_draft_amd.engine import EngineConfig, SpecEngine
    from vllm_omni_draft_amd.scheduler import Request, SpecDecodeScheduler
    from vllm_omni_draft_amd.spec_decode_worker import SpecDecodeConfig, SpecDecodeWorker
    rnd = random.Random(9)
    tw, dw = _weights(TARGET, 50), _weights(DRAFT, 51)
    prompts = [[rnd.randrange(TARGET.vocab) for _ in range(rnd.randint(2, 50))] for _ in range(7)]
    solo = SpecEngine(tw, None, EngineConfig(k=0, max_batch=1, max_seq_len=100, split_tokens=128))
    want = []
    for p in prompts:
        solo.reset()
        out = [solo.add_request(0, p)]
        while len(out) < 30:
            out.extend(solo.step()[0])
        want.append(out[:30])
    del solo
    w = SpecDecodeWorker(SpecDecodeConfig(target=tw, draft=dw, num_speculative_tokens=3, max_num_seqs=3,
                                          max_model_len=100, split_tokens=128))
    w.init_device()
    w.load_model()
    sched = SpecDecodeScheduler.for_worker(w)
    budgets = [5, 30, 1, 17, 30, 8, 22]
    eos = {1: want[1][6], 4: want[4][2]}
    reqs = [Request(f"q{i}", p, max_tokens=budgets[i], eos_token_id=eos.get(i)) for i, p in enumerate(prompts)]
    for r in reqs:
        sched.add_request(r)
    while sched.has_unfinished():
        sched.step()
        assert w.engine.active <= 3
    sched.drain()
    assert w.engine.active == 0
    for i, r in enumerate(reqs):
        exp = want[i][:budgets[i]]
        if i in eos:
            exp = exp[:exp.index(eos[i]) + 1]
            assert r.finish_reason == ("stop" if len(exp) < budgets[i] or exp[-1] == eos[i] else "length")
        assert r.output_token_ids == exp, i
    # a second wave reuses the freed slots and pages
    again = sched.generate(prompts[:4], max_tokens=6)
    assert [r.output_token_ids for r in again] == [wnt[:6] for wnt in want[:4]]


@pytest.mark.parametrize("rows", ["narrow", "wide"])
def test_scheduled_streams_match_the_oracle_golden(rows):
    """The caller side of the hot path against the ORACLE, not against another run of the
    same kernels: the golden prompts (tests/golden/e2e_tiny.json, written by the CPU oracle)
    go through SpecDecodeScheduler with 3 slots and staggered arrival (requests wait, join
    running batches, leave, slots are reused); every stream must equal the oracle's tokens up
    to the oracle's first near-tie."""
    import json
    import os
    from oracle.models import OracleLM
    from tests.test_gpu_model import _bf, _product
    from vllm_omni_draft_amd.scheduler import Request, SpecDecodeScheduler
    from vllm_omni_draft_amd.spec_decode_worker import SpecDecodeConfig, SpecDecodeWorker
    with open(os.path.join(os.path.dirname(__file__), "golden", "e2e_tiny.json")) as f:
        g = json.load(f)
    tcfg, dcfg = OracleConfig(**g["target"]), OracleConfig(**g["draft"])
    ts, ds = _bf(random_state(tcfg, g["target_seed"])), _bf(random_state(dcfg, g["draft_seed"]))
    w = SpecDecodeWorker(SpecDecodeConfig(target=_product(tcfg, ts), draft=_product(dcfg, ds),
                                          num_speculative_tokens=g["k"], max_num_seqs=3,
                                          max_model_len=128, split_tokens=128, rows=rows))
    w.init_device()
    w.load_model()
    sched = SpecDecodeScheduler.for_worker(w)
    reqs = [Request(f"g{i}", list(c["prompt"]), max_tokens=len(c["tokens"]))
            for i, c in enumerate(g["cases"])]
    arrivals = {0: [0, 1], 1: [2, 3], 3: [4], 6: [5, 6, 7]}       # step -> requests that arrive
    step = 0
    while sched.has_unfinished() or step <= max(arrivals):
        for i in arrivals.get(step, []):
            sched.add_request(reqs[i])
        sched.step()
        step += 1
        assert step < 400
    sched.drain()
    assert w.engine.active == 0
    lm = OracleLM(tcfg, ts, bf16_acts=True)
    n_tokens = n_match = 0
    for r, case in zip(reqs, g["cases"]):
        assert r.finish_reason == "length" and len(r.output_token_ids) == len(case["tokens"])
        seq, cache = list(case["prompt"]), lm.new_cache()
        logits = lm.forward(seq, 0, cache)[-1]
        for i, want in enumerate(case["tokens"]):
            if r.output_token_ids[i] != want:      # only legitimate at a near-tie of the oracle
                top2 = torch.topk(logits, 2).values
                margin = (top2[0] - top2[1]).item() / logits.abs().max().item()
                assert margin < 0.03, (r.request_id, i, r.output_token_ids[i], want, margin)
                break
            n_match += 1
            logits = lm.forward([want], len(seq), cache)[-1]
            seq.append(want)
        n_tokens += len(case["tokens"])
    assert n_match >= 0.6 * n_tokens, (n_match, n_tokens)


@pytest.mark.parametrize("seed,use_graph,rows,temperature", [
    (0, True, "wide", 0.0), (1, True, "wide", 0.0), (2, False, "wide", 0.0), (3, True, "wide", 0.0),
    (3, True, "narrow", 0.0), (4, False, "narrow", 0.0), (5, True, "narrow", 0.8), (6, False, "wide", 0.8),
    # (seeds whose schedule resumes a request in the step right after the one that preempted it,
    # with no step in between that leaves it out: the scheduler's flag alone tells the worker)
    (105, True, "narrow", 0.0), (121, False, "wide", 0.0)])
def test_random_v1_schedules_with_pieces_and_preemption(seed, use_graph, rows, temperature,
                                                        preempt_sampling=False):
    """A random V1 scheduler over the worker: prompts arrive whole or in pieces of random sizes,
    running requests are preempted (left out of steps) and return whole or in pieces, requests
    finish at random, also while they are parked or half prefilled; some ask for
    log-probabilities.  Every request's stream is the autoregressive stream it gets alone, and
    every reported entry belongs to the emitted token with rank 1 (greedy), across preemptions
    bit for bit on both kernel families: an engine that decodes on the <= 16-row family rebuilds a
    returning request's prompt with the prefill (wide) kernels and its emitted tokens with
    <= 16-row passes, as they were first computed.  temperature > 0: every request samples from its own seeded stream;
    the stream it gets alone (from a speculative engine of its own) is the stream it gets in the
    schedule, which then has no preemption (a returning request draws afresh).
    preempt_sampling (soak runs): preempt sampling requests too; streams are then not compared,
    the run checks that every step is served and the worker's books stay right."""
    from tests.test_host_logic import _NewRequestData, _sched
    from vllm_omni_draft_amd.engine import EngineConfig, SpecEngine
    from vllm_omni_draft_amd.spec_decode_worker import SpecDecodeConfig, SpecDecodeWorker
    rnd = random.Random(100 + seed)
    tw, dw = _weights(TARGET, 50), _weights(DRAFT, 51)
    n_req, n_new, slots = 10, 24, 3
    prompts = {f"q{i}": [rnd.randrange(TARGET.vocab) for _ in range(rnd.randint(1, 60))] for i in range(n_req)}
    sampling = temperature > 0.0
    solo = SpecEngine(tw, dw if sampling else None,
                      EngineConfig(k=3 if sampling else 0, max_batch=1, max_seq_len=140, split_tokens=128,
                                   rows=rows, temperature=temperature))
    want = {}
    for i, (r, p) in enumerate(prompts.items()):
        solo.reset()
        out = [solo.add_request(0, p, **({"seed": 1000 + i} if sampling else {}))]
        while len(out) < n_new:
            out.extend(solo.step()[0])
        want[r] = out[:n_new]
    del solo
    w = SpecDecodeWorker(SpecDecodeConfig(target=tw, draft=dw, num_speculative_tokens=3, max_num_seqs=slots,
                                          max_model_len=140, split_tokens=128, use_graph=use_graph, rows=rows,
                                          temperature=temperature))
    w.init_device()
    w.load_model()
    asks = {r: rnd.choice([None, None, 0, 2]) for r in prompts}
    waiting = list(prompts)
    got = {r: [] for r in prompts}
    left = {}            # request -> tokens still to prefill (new or returning)
    known = set()        # requests the worker has seen
    active, parked, done = [], [], set()
    steps = 0
    while len(done) < n_req:
        steps += 1
        assert steps < 3000
        new, cached, pieces, finished, resumed = [], [], {}, [], set()
        # requests that reached their length finish (wherever they are)
        for r in list(active) + list(parked):
            if len(got[r]) >= n_new:
                finished.append(r)
                (active if r in active else parked).remove(r)
                left.pop(r, None)
                done.add(r)
        # a random request is dropped early now and then, even half prefilled or parked
        pool = [r for r in active + parked if r not in finished]
        if pool and rnd.random() < 0.04:
            r = rnd.choice(pool)
            finished.append(r)
            (active if r in active else parked).remove(r)
            left.pop(r, None)
            done.add(r)
            want[r] = want[r][:len(got[r])]
        # return parked requests and admit new ones into the free slots
        returning = list(parked)         # (a request preempted in this step returns in a later one)
        # preempt
        for r in list(active):
            if r not in left and (not sampling or preempt_sampling) and rnd.random() < 0.07:
                active.remove(r)
                parked.append(r)
        while len(active) < slots and (returning or waiting) and rnd.random() < 0.7:
            if returning and (not waiting or rnd.random() < 0.5):
                r = returning.pop(rnd.randrange(len(returning)))
                parked.remove(r)
                resumed.add(r)       # recomputed from position 0 (resumed_from_preemption)
                left[r] = len(prompts[r]) + len(got[r])
            else:
                r = waiting.pop(0)
                left[r] = len(prompts[r])
            active.append(r)
        for r in active:
            if r in left:
                n = left[r] if rnd.random() < 0.5 else rnd.randint(1, left[r])
                pieces[r] = n
                left[r] -= n
                if left[r] == 0:
                    del left[r]
            if r in known:
                cached.append(r)
            else:
                known.add(r)
                sp = {"temperature": temperature}
                if sampling:
                    sp["seed"] = 1000 + int(r[1:])
                if asks[r] is not None:
                    sp["logprobs"] = asks[r]
                new.append(_NewRequestData(r, prompts[r], sampling_params=sp))
        rnd.shuffle(cached)
        if not (new or cached or finished):
            continue
        so = _sched(new=new, cached=cached, finished=[r for r in finished if r in known],
                    resumed=resumed)
        so.num_scheduled_tokens.update(pieces)
        out = w.execute_model(so)
        lp = out.logprobs
        for i, (r, toks) in enumerate(zip(out.req_ids, out.sampled_token_ids)):
            assert (toks == []) == (r in left), (r, toks)       # only finished prefills emit
            for j, t in enumerate(toks):
                if lp is not None:
                    n = lp.cu_num_generated_tokens[i] + j
                    assert lp.logprob_token_ids[n][0] == t and (sampling or lp.sampled_token_ranks[n] == 1)
            got[r].extend(toks)
        assert w.engine.active + len(w._prefill_of) == len(active)
        assert w._parked == set(parked)
    for r in prompts:
        n = min(n_new, len(want[r]))
        if preempt_sampling:
            assert all(0 <= t < TARGET.vocab for t in got[r]) and len(got[r]) >= n
        else:
            assert got[r][:n] == want[r][:n], r
    w.execute_model(_sched(finished=[r for r in known]))
    assert w.engine.active == 0 and not w._tokens and not w._parked and not w._prefill_of
