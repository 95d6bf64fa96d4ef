"""Launches the verify attention kernel (32 sequences x 4096 cached tokens, Llama-3-8B head
shape, 6 query tokens, 4096 keys per workgroup) 8 times over distinct caches, for counter
collection:

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- python3 experiments/pmc_attention.py
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d out -- python3 experiments/pmc_attention.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_omni_draft_amd import ops  # noqa: E402

B, ctx, Hq, Hkv, D, q_len, split, L = 32, 4096, 32, 8, 128, 6, 4096, 8
dev = "cuda"
pages_per_seq = (ctx + q_len + 15) // 16
n_pages = B * pages_per_seq
caches = [((torch.randn(n_pages, Hkv, D * 16, device=dev) * 0.5).to(torch.bfloat16),
           (torch.randn(n_pages, Hkv, D * 16, device=dev) * 0.5).to(torch.bfloat16)) for _ in range(L)]
bt = torch.randperm(n_pages, device=dev).to(torch.int32).view(B, pages_per_seq).contiguous()
q = torch.randn(B * q_len, Hq * D, device=dev).to(torch.bfloat16)
sl = torch.full((B,), ctx, dtype=torch.int32, device=dev)
ws = torch.empty(ops.paged_attention_workspace_bytes(B, q_len, Hq, D, ctx + q_len, split),
                 dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
for kc, vc in caches:
    ops.paged_attention(q, kc, vc, bt, sl, 0, B, q_len, Hq, Hkv, D, D ** -0.5, ctx + q_len, split,
                        workspace=ws)
torch.cuda.synchronize()
print("KV bytes per launch:", B * (ctx + q_len) * Hkv * D * 2 * 2)
