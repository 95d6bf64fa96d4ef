"""Soak run of the randomised engine and V1-schedule tests over seeds the test suite does not use:
    python experiments/soak_fuzz.py [first_seed] [n_seeds]
"""
import sys, itertools, traceback
sys.path.insert(0, ".")
import tests.test_gpu_fuzz as fz
import tests.test_gpu_tree as tt
bad = 0
n = 0
FIRST = int(sys.argv[1]) if len(sys.argv) > 1 else 100
COUNT = int(sys.argv[2]) if len(sys.argv) > 2 else 24
for seed in range(FIRST, FIRST + COUNT):
    for rows, graph, temp in (("narrow", True, 0.0), ("wide", False, 0.0), ("narrow", False, 0.8), ("wide", True, 0.8)):
        try:
            fz.test_random_v1_schedules_with_pieces_and_preemption(seed, graph, rows, temp)
            n += 1
        except Exception:
            bad += 1
            print("FAIL v1", seed, rows, graph, temp, flush=True)
            traceback.print_exc()
    for rows, graph in (("narrow", True), ("wide", False)):      # sampling requests preempted too
        try:
            fz.test_random_v1_schedules_with_pieces_and_preemption(seed, graph, rows, 0.8, preempt_sampling=True)
            n += 1
        except Exception:
            bad += 1
            print("FAIL v1 sampling with preemption", seed, rows, graph, flush=True)
            traceback.print_exc()
    for rows, mb, k, graph in (("narrow", 3, 3, True), ("wide", 6, 5, False), ("auto", 4, 1, True)):
        try:
            fz.test_random_join_leave_schedule(rows, mb, k, graph, seed)
            n += 1
        except Exception:
            bad += 1
            print("FAIL join", seed, rows, mb, k, graph, flush=True)
            traceback.print_exc()
    try:
        tt.test_draft_grown_trees_with_sequences_joining_and_leaving(seed)
        n += 1
    except Exception:
        bad += 1
        print("FAIL draft tree", seed, flush=True)
        traceback.print_exc()
    print("seed", seed, "done", n, "ok", bad, "bad", flush=True)
print("schedules ok", n, "bad", bad)


def scheduler_case(seed):
    """SpecDecodeScheduler over a real worker with random prompts, budgets, end-of-sequence tokens,
    slot counts and arrivals in between steps: every output is the solo stream cut at its stop."""
    import random
    from vllm_omni_draft_amd.engine import EngineConfig, SpecEngine
    from vllm_omni_draft_amd.scheduler import Request, SpecDecodeScheduler
    from vllm_omni_draft_amd.spec_decode_worker import SpecDecodeConfig, SpecDecodeWorker
    rnd = random.Random(seed)
    tw, dw = fz._weights(fz.TARGET, 50), fz._weights(fz.DRAFT, 51)
    n, slots, k = rnd.randint(3, 9), rnd.randint(1, 4), rnd.randint(1, 5)
    prompts = [[rnd.randrange(fz.TARGET.vocab) for _ in range(rnd.randint(1, 50))] for _ in range(n)]
    # (the worker picks its kernel family by its largest pass; streams compare within a family)
    rows = "wide" if slots * (k + 1) > 16 else "narrow"
    solo = SpecEngine(tw, None, EngineConfig(k=0, max_batch=1, max_seq_len=100, split_tokens=128, rows=rows))
    want = []
    for p in prompts:
        solo.reset()
        out = [solo.add_request(0, p)]
        while len(out) < 30:
            out.extend(solo.step()[0])
        want.append(out[:30])
    del solo
    w = SpecDecodeWorker(SpecDecodeConfig(target=tw, draft=dw, num_speculative_tokens=k, max_num_seqs=slots,
                                          max_model_len=100, split_tokens=128, use_graph=bool(seed % 2), rows=rows))
    w.init_device()
    w.load_model()
    sched = SpecDecodeScheduler.for_worker(w)
    budgets = [rnd.randint(1, 30) for _ in range(n)]
    eos = {i: want[i][rnd.randrange(0, 12)] for i in range(n) if rnd.random() < 0.3}
    reqs = [Request(f"q{i}", p, max_tokens=budgets[i], eos_token_id=eos.get(i)) for i, p in enumerate(prompts)]
    pending = list(reqs)
    for _ in range(rnd.randint(1, len(pending))):
        sched.add_request(pending.pop(0))
    while sched.has_unfinished() or pending:
        if pending and rnd.random() < 0.4:
            sched.add_request(pending.pop(0))
        sched.step()
        assert w.engine.active <= slots
    sched.drain()
    assert w.engine.active == 0
    for i, r in enumerate(reqs):
        exp = want[i][:budgets[i]]
        if i in eos and eos[i] in exp:
            exp = exp[:exp.index(eos[i]) + 1]
        assert r.output_token_ids == exp, (seed, i)


for seed in range(4 * FIRST, 4 * FIRST + 3 * COUNT):
    try:
        scheduler_case(seed)
        n += 1
    except Exception:
        bad += 1
        print("FAIL scheduler", seed, flush=True)
        traceback.print_exc()
print("TOTAL with scheduler cases ok", n, "bad", bad)
