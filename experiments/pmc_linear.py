"""Launches the dominant kernel of the step (norm + gate/up linear of the target
verify pass, M = 6) 32 times over distinct weight matrices, for counter
collection:

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- python3 experiments/pmc_linear.py
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d out -- python3 experiments/pmc_linear.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_omni_draft_amd import ops  # noqa: E402

H, F, M, L = 4096, 14336, 6, 32
gen = torch.Generator(device="cuda").manual_seed(0)
ws = [ops.tile_weights((torch.randn(2 * F, H, generator=gen, device="cuda") * 0.02).to(torch.bfloat16),
                       ops.TILE_GATED) for _ in range(L)]
g = torch.ones(H, dtype=torch.bfloat16, device="cuda")
x = torch.randn(M, H, generator=gen, device="cuda").to(torch.bfloat16)
out = torch.empty(M, F, dtype=torch.bfloat16, device="cuda")
torch.cuda.synchronize()
for w in ws:
    ops.norm_linear(w, x, g, None, 1e-5, False, None, ops.EPI_SILU_MUL, out=out)
torch.cuda.synchronize()
print("bytes per launch:", ws[0].nbytes() + x.numel() * 2 + out.numel() * 2)
