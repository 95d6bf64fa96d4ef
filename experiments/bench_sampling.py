"""Step time of the sampling path (temperature 1) next to the greedy path, same models.
    python experiments/bench_sampling.py
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_omni_draft_amd import config, weights  # noqa: E402
from vllm_omni_draft_amd.engine import EngineConfig, SpecEngine  # noqa: E402

tw = weights.random_init(config.preset("llama-3-8b"), seed=0)
dw = weights.random_init(config.preset("llama-3.2-1b"), seed=1)
prompt = torch.randint(0, 128256, (1024,), generator=torch.Generator().manual_seed(1234)).tolist()
for temp in (0.0, 1.0):
    eng = SpecEngine(tw, dw, EngineConfig(k=5, max_seq_len=1024 + 600, temperature=temp, seed=3))
    eng.add_request(0, prompt)
    for _ in range(8):
        eng.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    for _ in range(48):
        n += len(eng.step()[0])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    m = eng.metrics()
    print(f"temperature {temp}: {dt / 48 * 1e3:.3f} ms/step, {n / 48:.2f} tokens/step, "
          f"accepted/draft = {m['accepted']}/{m['draft']}", flush=True)
    del eng
