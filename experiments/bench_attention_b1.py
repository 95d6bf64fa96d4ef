"""Latency of the attention launch (+ merge) at batch 1 inside a replayed hipGraph:
a chain of L launches over L distinct KV caches.
    python experiments/bench_attention_b1.py
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


def graph_time(enqueue, reps=7):
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


def case(B, ctx, Hq, Hkv, D, q_len, split, max_ctx, L=32):
    pages_per_seq = (max_ctx + 15) // 16
    n_pages = B * pages_per_seq
    kcs = [(torch.randn(n_pages, Hkv, D * 16, device=dev) * 0.5).to(torch.bfloat16) for _ in range(L)]
    vcs = [(torch.randn(n_pages, Hkv, D * 16, device=dev) * 0.5).to(torch.bfloat16) for _ in range(L)]
    bt = torch.randperm(n_pages, device=dev).to(torch.int32).view(B, pages_per_seq).contiguous()
    q = torch.randn(B * q_len, Hq * D, device=dev).to(torch.bfloat16)
    sl = torch.full((B,), ctx, dtype=torch.int32, device=dev)
    ws = torch.empty(ops.paged_attention_workspace_bytes(B, q_len, Hq, D, max_ctx, split),
                     dtype=torch.uint8, device=dev)
    out = None

    def enqueue():
        for kc, vc in zip(kcs, vcs):
            ops.paged_attention(q, kc, vc, bt, sl, 0, B, q_len, Hq, Hkv, D, D ** -0.5, max_ctx,
                                split, workspace=ws)
    us = graph_time(enqueue) / L
    print(f"B={B} ctx={ctx:5d} max_ctx={max_ctx:5d} Hq={Hq} Hkv={Hkv} D={D:3d} q_len={q_len} "
          f"split={split:5d}: {us:6.2f} us per attention(+merge)", flush=True)


if __name__ == "__main__":
    for split in (128, 256, 512, 2048):
        case(1, 1100, 32, 8, 64, 1, split, 1280)
    for split in (128, 256, 512, 2048):
        case(1, 1100, 32, 8, 128, 6, split, 1280)
    case(1, 1100, 32, 8, 128, 1, 256, 1280)
    case(1, 200, 32, 8, 64, 1, 256, 256)
    case(1, 200, 32, 8, 128, 6, 256, 256)
    case(8, 1100, 32, 8, 128, 6, 256, 1280)
