"""Step time of draft-grown trees next to the chain step, headline models (Llama-3-8B target,
Llama-3.2-1B draft, bf16, batch 1, 1024-token prompt).  Random weights accept nothing, so this
times the STEP only (all passes run whatever is accepted); what a tree buys in accepted tokens
depends on the draft's top-b calibration and is not measured here.
    python experiments/bench_draft_tree.py [steps]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_omni_draft_amd import config, weights  # noqa: E402
from vllm_omni_draft_amd.engine import EngineConfig, SpecEngine  # noqa: E402

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 48
tcfg = config.preset(os.environ.get("TARGET", "llama-3-8b"))
tw = weights.random_init(tcfg, seed=0)
dw = weights.random_init(config.preset(os.environ.get("DRAFT", "llama-3.2-1b"), vocab=tcfg.vocab), seed=1)
prompt = torch.randint(0, tcfg.vocab, (1024,), generator=torch.Generator().manual_seed(0)).tolist()


def run(name, k, **kw):
    eng = SpecEngine(tw, dw if k else None, EngineConfig(k=k, max_batch=1, max_seq_len=1024 + 6 * (STEPS + 16),
                                                         split_tokens=256, **kw))
    eng.add_request(0, prompt)
    for _ in range(8):
        eng.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    for _ in range(STEPS):
        n += len(eng.step()[0])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / STEPS
    print(f"{name}: {dt * 1e3:.3f} ms/step, {n / STEPS:.2f} tokens/step (random weights)", flush=True)
    del eng
    torch.cuda.empty_cache()


run("autoregressive       ", 0)
run("chain k=3            ", 3)
run("chain k=5            ", 5)
for b, k in ((2, 3), (4, 3), (2, 5), (3, 5)):
    run(f"tree b={b} k={k} ({b * k:2d} nodes)", k, draft_tree_branches=b)
