"""Soak run at engine level over contexts that cross the context buckets of the step graphs and
the attention split boundaries while sequences join and leave: prompts of 1..400 tokens, 60 new
tokens each, random k and batch size, both kernel families, graphs on and off, chains and
draft-grown trees, tokens read back after every step, after several steps, or one step late
(overlapped).  Every stream must be the one the request gets alone.
    python experiments/soak_long.py [first_seed] [n_seeds]
"""
import random
import sys
import traceback

sys.path.insert(0, ".")
import tests.test_gpu_fuzz as fz  # noqa: E402
from vllm_omni_draft_amd.engine import EngineConfig, SpecEngine  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 30
tw, dw = fz._weights(fz.TARGET, 50), fz._weights(fz.DRAFT, 51)
N_NEW = 60


def case(seed):
    rnd = random.Random(seed)
    rows = rnd.choice(["narrow", "wide"])
    slots, k = rnd.randint(1, 3), rnd.randint(1, 3)
    kw = {}
    if rnd.random() < 0.4:
        kw["draft_tree_branches"] = rnd.randint(2, 3)
    split = rnd.choice([128, 256])
    n_req = rnd.randint(slots, slots + 3)
    prompts = [[rnd.randrange(fz.TARGET.vocab) for _ in range(rnd.choice([1, 100, 127, 128, 129, 250, 255, 256,
                                                                          257, rnd.randint(1, 400)]))]
               for _ in range(n_req)]
    solo = SpecEngine(tw, None, EngineConfig(k=0, max_batch=1, max_seq_len=480, split_tokens=split, rows=rows))
    want = []
    for p in prompts:
        solo.reset()
        out = [solo.add_request(0, p)]
        while len(out) < N_NEW:
            out.extend(solo.step()[0])
        want.append(out[:N_NEW])
    del solo
    eng = SpecEngine(tw, dw, EngineConfig(k=k, max_batch=slots, max_seq_len=500, split_tokens=split, rows=rows,
                                          use_graph=rnd.random() < 0.7, **kw))
    mode = rnd.choice(["sync", "sync", "overlapped", "several"])   # how the host takes the tokens

    def advance():
        if mode == "sync":
            return [eng.step()]
        if mode == "several":        # a few steps without a read-back in between
            for _ in range(rnd.randint(1, 3)):
                eng.step(sync=False)
            return [eng.read_back()]
        got_now = []                 # overlapped: tokens arrive one call late; flush before the batch changes
        for _ in range(rnd.randint(1, 4)):
            o = eng.step_overlapped()
            if o:
                got_now.append(o)
        got_now.append(eng.flush())
        return got_now

    live, waiting, got = {}, list(range(n_req)), {}
    while live or waiting:
        while waiting and len(live) < slots and (not live or rnd.random() < 0.5):
            j = waiting.pop(0)
            s = eng.active
            live[s] = [j, [eng.add_request(s, prompts[j])]]
        for outs in advance():
            for s in sorted(live):
                live[s][1].extend(outs[s])
        for s in sorted(live, reverse=True):
            j, toks = live[s]
            if len(toks) >= N_NEW:
                got[j] = toks[:N_NEW]
                last = max(live)
                eng.remove_request(s)
                if s != last:
                    live[s] = live[last]
                del live[last]
    for j in range(n_req):
        assert got[j] == want[j], (seed, j, rows, slots, k, kw, split, mode, len(prompts[j]))


ok = bad = 0
for seed in range(first, first + count):
    try:
        case(seed)
        ok += 1
    except Exception:
        bad += 1
        print("FAIL", seed, flush=True)
        traceback.print_exc()
    print("seed", seed, "ok", ok, "bad", bad, flush=True)
print("TOTAL ok", ok, "bad", bad)
