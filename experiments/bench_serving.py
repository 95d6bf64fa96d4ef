"""Continuous batching end to end (config 3's shape): N requests arrive at t0 with ragged prompts
(64..1024 tokens, seed 7), each generates 256 tokens, at most 32 run at once; the scheduler admits
waiting requests as slots free up.  Output tokens/s of the whole job, prefill included:
autoregressive, speculative, and speculative under the engine's own policy (speculative_policy=
"auto": per step, whichever of the two kinds of step has been yielding more tokens per second at
the current batch size); THRESHOLDS=1 adds fixed batch thresholds.  Random
weights: as in the other benches every proposal is replaced by the target's own token with
probability 0.8 (the autoregressive run provides each request's stream).
    python experiments/bench_serving.py [n_requests] [max_num_seqs] [max_tokens]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_omni_draft_amd import config, weights  # noqa: E402
from vllm_omni_draft_amd.scheduler import Request, SpecDecodeScheduler  # noqa: E402
from vllm_omni_draft_amd.spec_decode_worker import SpecDecodeConfig, SpecDecodeWorker  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S = int(sys.argv[2]) if len(sys.argv) > 2 else 32
T = int(sys.argv[3]) if len(sys.argv) > 3 else 256
TARGET = os.environ.get("TARGET", "qwen2.5-7b")
DRAFT = os.environ.get("DRAFT", "qwen2.5-0.5b")
tcfg = config.preset(TARGET)
tw = weights.random_init(tcfg, seed=0)
dw = weights.random_init(config.preset(DRAFT, vocab=tcfg.vocab), seed=1)
g = torch.Generator().manual_seed(7)
lens = [int(n) for n in torch.randint(64, 1025, (N,), generator=g)]
prompts = [torch.randint(0, tcfg.vocab, (n,), generator=g).tolist() for n in lens]


class Worker(SpecDecodeWorker):
    """Hands every admitted request's known stream to the engine's acceptance control."""
    streams = {}

    def _multimodal_inputs(self, g):
        row = self.streams.get(g.request_id)
        return {} if row is None else {"truth_row": row}


def run(name, k, disable=0, truth=None, policy="always"):
    w = Worker(SpecDecodeConfig(target=tw, draft=dw if k else None, num_speculative_tokens=k,
                                          max_num_seqs=S, max_model_len=1024 + T + 8,
                                          speculative_disable_by_batch_size=disable,
                                          speculative_policy=policy,
                                          rows="wide" if S > 2 else "auto"))
    w.init_device()
    w.load_model()
    sched = SpecDecodeScheduler.for_worker(w)
    sched.generate(prompts[:2], max_tokens=8)          # warm-up: graphs of the common shapes
    Worker.streams = {}
    if truth is not None:
        table = torch.zeros(S, 1024 + T + 16, dtype=torch.int32, device="cuda")
        w.engine.set_synthetic_acceptance(table, 0.8, seed=3)
        Worker.streams = {f"r{i}": prompts[i] + truth[i] for i in range(N)}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reqs = [Request(f"r{i}", p, max_tokens=T) for i, p in enumerate(prompts)]
    for r in reqs:
        sched.add_request(r)
    steps0 = sched.steps
    while sched.has_unfinished():
        sched.step()
    sched.drain()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = sum(len(r.output_token_ids) for r in reqs)
    m = w.engine.metrics()
    acc = m["accepted"] / max(m["draft"], 1)
    print(f"{name}: {n} tokens in {dt:.2f} s = {n / dt:.0f} tok/s, {sched.steps - steps0} steps, "
          f"draft acceptance {acc:.2f}", flush=True)
    if policy == "auto":
        for B, st in sorted(w.engine.policy_state().items()):
            if st["t_spec"] and st["t_ar"]:
                print(f"    B={B}: spec {st['t_spec'] * 1e3:.2f} ms x {st['emitted'] or 0:.2f} tok, "
                      f"plain {st['t_ar'] * 1e3:.2f} ms -> {st['mode']}", flush=True)
    out = [r.output_token_ids for r in reqs]
    del sched, w
    torch.cuda.empty_cache()
    return out


base = run("autoregressive      ", 0)
spec = run("speculative k=5     ", 5, truth=base)
auto = run("k=5, policy auto   ", 5, truth=base, policy="auto")
print("streams equal:", base == spec == auto)
if os.environ.get("THRESHOLDS"):
    half = run("k=5, off above 8 seq", 5, disable=8, truth=base)
    mid = run("k=5, off above 16   ", 5, disable=16, truth=base)
    print("streams equal:", base == half == mid)
