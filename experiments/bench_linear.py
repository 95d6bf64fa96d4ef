"""Per-kernel time of the weight-streaming linears inside a replayed hipGraph
(the regime the engine runs them in): a chain of L launches over L distinct
weight matrices, timed with events around the graph launch.

    python experiments/bench_linear.py
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_omni_draft_amd import _lib, ops  # noqa: E402

dev = "cuda"
lib = _lib.load()
stream = torch.cuda.Stream()


def graph_time(enqueue, reps=5):
    torch.cuda.synchronize()  # inputs were filled on the default stream
    with torch.cuda.stream(stream):
        enqueue()
        torch.cuda.synchronize()
        g = C.c_void_p()
        _lib.check(lib.sd_graph_begin(stream.cuda_stream), "begin")
        enqueue()
        _lib.check(lib.sd_graph_end(stream.cuda_stream, C.byref(g)), "end")
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            lib.sd_graph_launch(g, stream.cuda_stream)
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        lib.sd_graph_destroy(g)
    ts.sort()
    return ts[len(ts) // 2]


def case(name, N, K, M, kind, L=48):
    gen = torch.Generator(device=dev).manual_seed(0)
    rows = 2 * N if kind == "gated" else N
    mode = {"gated": ops.TILE_GATED, "qkv": ops.TILE_ROTARY}.get(kind, ops.TILE_PLAIN)
    ws = [ops.tile_weights((torch.randn(rows, K, generator=gen, device=dev) * 0.02).to(torch.bfloat16),
                           mode, 64 if kind == "qkv" else 0) for _ in range(L)]
    x = torch.randn(M, K, generator=gen, device=dev).to(torch.bfloat16)
    g = torch.ones(K, dtype=torch.bfloat16, device=dev)
    res = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    if kind == "gated":
        fn = lambda w: ops.norm_linear(w, x, g, None, 1e-5, False, None, ops.EPI_SILU_MUL, out=out)
    elif kind == "norm":
        fn = lambda w: ops.norm_linear(w, x, g, None, 1e-5, False, None, ops.EPI_STORE, out=out)
    elif kind == "res":
        fn = lambda w: ops.linear(w, x, None, res, ops.EPI_RESIDUAL, out=out)
    else:
        fn = lambda w: ops.linear(w, x, None, None, ops.EPI_STORE, out=out)
    us = graph_time(lambda: [fn(w) for w in ws]) / L
    mb = ws[0].nbytes() / 1e6
    print(f"{name:28s} N={N:6d} K={K:6d} M={M:2d} {mb:7.1f} MB  {us:7.2f} us/kernel  {mb / us:6.2f} TB/s",
          flush=True)


def empty_chain(L=64):
    x = torch.zeros(1, 32, dtype=torch.bfloat16, device=dev)
    w = ops.tile_weights(torch.zeros(16, 32, dtype=torch.bfloat16, device=dev))
    out = torch.empty(1, 16, dtype=torch.bfloat16, device=dev)
    us = graph_time(lambda: [ops.linear(w, x, None, None, ops.EPI_STORE, out=out) for _ in range(L)]) / L
    print(f"trivial 1-tile linear kernel: {us:.2f} us/kernel (boundary + fixed cost)")


if __name__ == "__main__":
    empty_chain()
    for M in (1, 6):
        print(f"--- draft shapes (Llama-3.2-1B), M={M}")
        case("qkv (norm)", 3072, 2048, M, "norm")
        case("o (+res)", 2048, 2048, M, "res")
        case("gate/up (norm, silu)", 8192, 2048, M, "gated")
        case("down (+res)", 2048, 8192, M, "res")
        print(f"--- target shapes (Llama-3-8B), M={M}")
        case("qkv (norm)", 6144, 4096, M, "norm")
        case("o (+res)", 4096, 4096, M, "res")
        case("gate/up (norm, silu)", 14336, 4096, M, "gated", L=24)
        case("down (+res)", 4096, 14336, M, "res", L=24)
    case("lm_head 1B (norm,f32~store)", 128256, 2048, 1, "norm", L=8)
