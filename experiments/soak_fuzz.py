"""Soak run of the randomised engine and V1-schedule tests over seeds the test suite does not use:
    python experiments/soak_fuzz.py
"""
import sys, itertools, traceback
sys.path.insert(0, ".")
import tests.test_gpu_fuzz as fz
import tests.test_gpu_tree as tt
bad = 0
n = 0
for seed in range(100, 124):
    for rows, graph, temp in (("narrow", True, 0.0), ("wide", False, 0.0), ("narrow", False, 0.8), ("wide", True, 0.8)):
        try:
            fz.test_random_v1_schedules_with_pieces_and_preemption(seed, graph, rows, temp)
            n += 1
        except Exception:
            bad += 1
            print("FAIL v1", seed, rows, graph, temp, flush=True)
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
print("TOTAL ok", n, "bad", bad)
