"""Autoregressive step time of Llama-3-8B shapes at 8 and 16 sequences on the two kernel families
(the narrow one takes up to 16 rows per pass).
    python experiments/bench_ar_family.py
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_omni_draft_amd import config, weights
from vllm_omni_draft_amd.engine import EngineConfig, SpecEngine
tw = weights.random_init(config.preset("llama-3-8b"), seed=0)
g = torch.Generator().manual_seed(7)
for B in (8, 16):
    prompts = [torch.randint(0, 128256, (1024,), generator=g).tolist() for _ in range(B)]
    for rows in ("narrow", "wide"):
        eng = SpecEngine(tw, None, EngineConfig(k=0, max_batch=B, max_seq_len=1400, rows=rows))
        for i, p in enumerate(prompts):
            eng.add_request(i, p)
        for _ in range(8):
            eng.step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100):
            eng.step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
        print(f"B={B} rows={rows}: {dt*1e3:.3f} ms/step, {B/dt:.0f} tok/s", flush=True)
        del eng; torch.cuda.empty_cache()
