"""Soak run of the sampling kernels against the oracle over random shapes (batch, proposals,
vocabulary sizes that are no round numbers, split and unsplit forms): rejection sampling, typical
acceptance, the top-k / top-p / min-p filter, the log-probability report.
    python experiments/soak_sampling.py [seconds]
"""
import random
import sys
import time
import traceback

sys.path.insert(0, ".")
import tests.test_gpu_sampling as ts  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 180.0
rnd = random.Random(77)
t_end = time.time() + budget
n_ok = n_bad = 0


def attempt(fn, *args):
    global n_ok, n_bad
    try:
        fn(*args)
        n_ok += 1
    except Exception:
        n_bad += 1
        print("FAIL", fn.__name__, args, flush=True)
        traceback.print_exc()


def vocab():
    return rnd.choice([64, 100, 1000, 4100, 32000, 50272, 128256, 151936, 4 * rnd.randint(16, 40000)])


while time.time() < t_end:
    attempt(ts.test_rejection_random_full_vocab, rnd.randint(1, 40), rnd.randint(1, 15), vocab(), rnd.random() < 0.5)
    attempt(ts.test_typical_acceptance_full_vocab, rnd.randint(1, 16), rnd.randint(1, 15), vocab(),
            rnd.choice([2.0, 4.0, 8.0]))
    attempt(ts.test_top_k_top_p_min_p_filter_matches_oracle, max(vocab(), 1000))
    attempt(ts.test_token_logprobs_match_oracle, max(vocab(), 1000))
    attempt(ts.test_softmax_and_sample_rows, vocab(), rnd.random() < 0.5)
    print("cases", n_ok + n_bad, "bad", n_bad, flush=True)
print("TOTAL ok", n_ok, "bad", n_bad)
