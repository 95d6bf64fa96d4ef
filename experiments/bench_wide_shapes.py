"""The two many-row linear kernels, A/B on one box (`make wide-switch` builds the library with
the run-time switch):
    SD_AMD_LIB=build/exp/libsd_amd_switch.so python experiments/bench_wide_shapes.py [--quick]
Per layer matrix of Llama-3-8B and Qwen2.5-7B and per row count: the eight-wave kernel (loads in
a burst ahead of a chunk's MFMAs) against the four-wave kernel (loads spread through the MFMA
stream, weights two chunks ahead), same groups, same bits.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_omni_draft_amd import _lib, ops  # noqa: E402

lib = _lib.load()
dev = "cuda"


def timed(fn, ws, reps=3):
    for w in ws:
        fn(w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        for w in ws:
            fn(w)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * len(ws))


def case(name, N, K, kind, Ms=(96, 128, 192, 256), L=6, fp8=False):
    gen = torch.Generator(device=dev).manual_seed(0)
    rows = 2 * N if kind == "gated" else N
    mode = ops.TILE_GATED if kind == "gated" else ops.TILE_PLAIN
    ws = []
    for _ in range(L):
        W = (torch.randn(rows, K, generator=gen, device=dev) * 0.02).to(torch.bfloat16)
        if fp8:
            q, s = ops.quantize_fp8(W)
            ws.append(ops.tile_weights(q, mode, 0, scale=s))
        else:
            ws.append(ops.tile_weights(W, mode, 0))
        del W
    epi = ops.EPI_SILU_MUL if kind == "gated" else (ops.EPI_RESIDUAL if kind == "res" else ops.EPI_STORE)
    for M in Ms:
        x = torch.randn(M, K, generator=gen, device=dev).to(torch.bfloat16)
        res = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
        outs = []
        ts = []
        for four in (0, -1):
            lib.sd_debug_wide_four_waves(four)
            out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            wsb = torch.empty(int(lib.sd_linear_wide_workspace_bytes(M, N, K, epi)) + 16, dtype=torch.uint8,
                              device=dev)
            r = res if kind == "res" else None
            ts.append(timed(lambda w: ops.linear_wide(w, x, None, r, epi, out=out, workspace=wsb), ws))
            outs.append(out.clone())
        same = torch.equal(outs[0], outs[1])
        fl = 2.0 * M * rows * K
        print(f"{name:14s}{' fp8' if fp8 else '    '} N={N:6d} K={K:6d} M={M:3d}: eight-wave kernel {ts[0]:7.1f} us | library's "
              f"choice {ts[1]:7.1f} us ({ts[0] / ts[1]:.2f}x, {fl / ts[1] * 1e-6:5.0f} TFLOP/s) bits equal: {same}", flush=True)
    lib.sd_debug_wide_four_waves(-1)


if __name__ == "__main__":
    if "--quick" in sys.argv:      # one row count (192: what experiments/build_wide_exp.sh builds)
        Ms = (int(sys.argv[sys.argv.index("--rows") + 1]) if "--rows" in sys.argv else 192,)
        for _ in range(2):
            case("llama gate/up", 14336, 4096, "gated", Ms=Ms)
        case("llama down", 4096, 14336, "res", Ms=Ms)
        case("qwen gate/up", 18944, 3584, "gated", Ms=Ms)
        case("llama head", 128256, 4096, "plain", Ms=Ms, L=2)
        case("qwen head", 152064, 3584, "plain", Ms=Ms, L=2)
        sys.exit(0)
    case("llama gate/up", 14336, 4096, "gated")
    case("llama qkv", 6144, 4096, "plain")
    case("llama o", 4096, 4096, "res")
    case("llama down", 4096, 14336, "res")
    case("qwen gate/up", 18944, 3584, "gated")
    case("qwen qkv", 4608, 3584, "plain")
    case("qwen o", 3584, 3584, "res")
    case("qwen down", 3584, 18944, "res")
    case("llama head", 128256, 4096, "plain", L=2)
    case("qwen head", 152064, 3584, "plain", L=2)
    case("llama gate/up", 14336, 4096, "gated", fp8=True)
    case("qwen down", 3584, 18944, "res", fp8=True)
