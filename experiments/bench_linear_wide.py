"""Time of the wide (17..256 token rows) linear over distinct weight matrices, eager launches.
    python experiments/bench_linear_wide.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_omni_draft_amd import ops  # noqa: E402

dev = "cuda"


def case(name, N, K, M, kind, L=12):
    gen = torch.Generator(device=dev).manual_seed(0)
    rows = 2 * N if kind == "gated" else N
    mode = ops.TILE_GATED if kind == "gated" else ops.TILE_PLAIN
    ws = [ops.tile_weights((torch.randn(rows, K, generator=gen, device=dev) * 0.02).to(torch.bfloat16), mode, 0)
          for _ in range(L)]
    x = torch.randn(M, K, generator=gen, device=dev).to(torch.bfloat16)
    res = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    epi = ops.EPI_SILU_MUL if kind == "gated" else (ops.EPI_RESIDUAL if kind == "res" else ops.EPI_STORE)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    wsb = torch.empty(int(ops._lib.load().sd_linear_wide_workspace_bytes(M, N, K, epi)) + 16,
                      dtype=torch.uint8, device=dev)
    fn = lambda w: ops.linear_wide(w, x, None, res if kind == "res" else None, epi, out=out, workspace=wsb)
    for w in ws:
        fn(w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        for w in ws:
            fn(w)
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (3 * L)
    mb = ws[0].nbytes() / 1e6
    print(f"{name:22s} N={N:6d} K={K:6d} M={M:3d} {mb:7.1f} MB  {us:8.2f} us  {mb / us:5.2f} TB/s "
          f"(weights)  ws {wsb.numel() / 1e6:.1f} MB", flush=True)


if __name__ == "__main__":
    for M in (32, 192):
        print(f"--- Qwen2.5-7B shapes, M={M}")
        case("qkv", 4608, 3584, M, "plain")
        case("o (+res)", 3584, 3584, M, "res")
        case("gate/up (silu)", 18944, 3584, M, "gated")
        case("down (+res)", 3584, 18944, M, "res")
    case("lm_head", 152064, 3584, 192, "plain", L=3)
    print("--- Qwen2.5-0.5B shapes, M=32")
    case("qkv", 1152, 896, 32, "plain")
    case("gate/up (silu)", 4864, 896, 32, "gated")
    case("down (+res)", 896, 4864, 32, "res")
