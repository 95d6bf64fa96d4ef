"""Prefill time of one 1024-token prompt (target + draft, 256-row passes of the wide kernel family).
    python experiments/bench_prefill.py [target] [draft] [prompt_len]
WIDE_FOUR=0 with the switch library (make wide-switch, SD_AMD_LIB): never the four-wave linear kernel.
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_omni_draft_amd import _lib, config, weights  # noqa: E402
from vllm_omni_draft_amd.engine import EngineConfig, SpecEngine  # noqa: E402

if os.environ.get("WIDE_FOUR"):
    _lib.load().sd_debug_wide_four_waves(int(os.environ["WIDE_FOUR"]))
TARGET = sys.argv[1] if len(sys.argv) > 1 else "llama-3-8b"
DRAFT = sys.argv[2] if len(sys.argv) > 2 else "llama-3.2-1b"
P = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
tcfg = config.preset(TARGET)
tw = weights.random_init(tcfg, seed=0)
dw = weights.random_init(config.preset(DRAFT, vocab=tcfg.vocab), seed=1)
eng = SpecEngine(tw, dw, EngineConfig(k=5, max_batch=1, max_seq_len=P + 64, split_tokens=256))
prompt = torch.randint(0, tcfg.vocab, (P,), generator=torch.Generator().manual_seed(0)).tolist()
ts = []
for _ in range(6):
    eng.reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.add_request(0, prompt)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
ts = sorted(ts[1:])
print(f"{TARGET} + {DRAFT}, prompt {P}: prefill {ts[len(ts) // 2] * 1e3:.2f} ms (median of 5), "
      f"{P / ts[len(ts) // 2]:.0f} prompt tokens/s", flush=True)
