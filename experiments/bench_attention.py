"""Verify-attention kernel alone at KV-bound sizes, several split sizes.
    python experiments/bench_attention.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_omni_draft_amd import ops  # noqa: E402

dev = "cuda"


def run(B, ctx, Hq, Hkv, D, q_len, split):
    pages_per_seq = (ctx + q_len + 15) // 16
    n_pages = B * pages_per_seq
    kc = (torch.randn(n_pages, Hkv, D * 16, device=dev) * 0.5).to(torch.bfloat16)
    vc = (torch.randn(n_pages, Hkv, D * 16, device=dev) * 0.5).to(torch.bfloat16)
    bt = torch.randperm(n_pages, device=dev).to(torch.int32).view(B, pages_per_seq).contiguous()
    q = torch.randn(B * q_len, Hq * D, device=dev).to(torch.bfloat16)
    sl = torch.full((B,), ctx, dtype=torch.int32, device=dev)
    ws = torch.empty(ops.paged_attention_workspace_bytes(B, q_len, Hq, D, ctx + q_len, split),
                     dtype=torch.uint8, device=dev)
    f = lambda: ops.paged_attention(q, kc, vc, bt, sl, 0, B, q_len, Hq, Hkv, D, D ** -0.5,
                                    ctx + q_len, split, workspace=ws)
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    kv = B * (ctx + q_len) * Hkv * D * 2 * 2
    print(f"B={B:3d} ctx={ctx:6d} Hq={Hq} Hkv={Hkv} D={D} q_len={q_len} split={split:5d}: "
          f"{us:8.1f} us  {kv / us * 1e-6:5.2f} TB/s  ({kv / 1e6:.0f} MB)", flush=True)


if __name__ == "__main__":
    for split in (256, 512, 1024, 2048, 4096):
        run(32, 4096, 32, 8, 128, 6, split)
    run(8, 16384, 32, 8, 128, 6, 2048)
    run(64, 2048, 32, 8, 128, 6, 1024)
    run(32, 4096, 32, 8, 128, 1, 1024)
    run(32, 4096, 32, 8, 64, 1, 1024)
    run(32, 4096, 28, 4, 128, 6, 1024)
    run(32, 4096, 64, 8, 128, 6, 1024)   # 48 rows per KV head: two 32-row MFMA tiles
