"""The draft layer's launches one by one, in replayed graphs over 16 layers of distinct weights
(Llama-3.2-1B widths, one row, ~1100 cached tokens): the five-launch layer's kernels against
sd_draft_attn_oproj / sd_draft_mlp.
    python experiments/bench_draft_layer.py [rows]
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
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1
L, H, F, Hq, Hkv, D, ctx, max_ctx, split = 16, 2048, 8192, 32, 8, 64, 1100, 2048, 256


def graph_time(enqueue, reps=9):
    torch.cuda.synchronize()
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
    return ts[len(ts) // 2] / L


def rnd(*shape, s=0.03):
    return (torch.randn(*shape, device=dev) * s).to(torch.bfloat16)


q_len = 1 if M == 1 else 2
B = M // q_len
pages = (max_ctx + 15) // 16
n_pages = B * pages
kcs = [rnd(n_pages, Hkv, D * 16, s=0.5) for _ in range(L)]
vcs = [rnd(n_pages, Hkv, D * 16, s=0.5) for _ in range(L)]
bt = torch.randperm(n_pages, device=dev).to(torch.int32).view(B, pages).contiguous()
sl = torch.full((B,), ctx, dtype=torch.int32, device=dev)
q = rnd(M, Hq * D, s=1.0)
o_w = [ops.tile_weights(rnd(H, Hq * D)) for _ in range(L)]
up_w = [ops.tile_weights(rnd(2 * F, H), ops.TILE_GATED) for _ in range(L)]
down_w = [ops.tile_weights(rnd(H, F)) for _ in range(L)]
down_km = [ops.kmajor_tiles(w) for w in down_w]
norm_w = (1 + 0.05 * torch.randn(H, device=dev)).to(torch.bfloat16)
h = rnd(M, H, s=1.0)
h2 = torch.empty_like(h)
act = torch.empty(M, F, dtype=torch.bfloat16, device=dev)
acc = torch.zeros(M, H, dtype=torch.float32, device=dev)
ws = torch.empty(ops.paged_attention_workspace_bytes(B, q_len, Hq, D, max_ctx, split), dtype=torch.uint8, device=dev)
slabs = torch.zeros(Hkv, M, H, dtype=torch.float32, device=dev)


def five_attn():
    for l in range(L):
        ops.paged_attention(q, kcs[l], vcs[l], bt, sl, 0, B, q_len, Hq, Hkv, D, D ** -0.5, max_ctx, split,
                            workspace=ws, partials_only=True)


def five_attn_oproj():
    for l in range(L):
        ops.paged_attention(q, kcs[l], vcs[l], bt, sl, 0, B, q_len, Hq, Hkv, D, D ** -0.5, max_ctx, split,
                            workspace=ws, partials_only=True)
        ops.linear_attn_merge(o_w[l], ws, sl, 0, B, q_len, Hq, Hkv, D, max_ctx, split, residual=h,
                              epilogue=ops.EPI_RESIDUAL, out=h2)


def three_attn_oproj():
    for l in range(L):
        lib.sd_draft_attn_oproj(q.data_ptr(), Hq * D, kcs[l].data_ptr(), vcs[l].data_ptr(), bt.data_ptr(), pages,
                                sl.data_ptr(), 0, o_w[l].data.data_ptr(), slabs.data_ptr(), acc.data_ptr(),
                                acc.numel(), B, q_len, Hq, Hkv, D, H, D ** -0.5, max_ctx, stream.cuda_stream)


def five_mlp():
    for l in range(L):
        ops.norm_linear(up_w[l], h, norm_w, None, 1e-5, False, None, ops.EPI_SILU_MUL, out=act)
        ops.linear(down_w[l], act, None, h, ops.EPI_RESIDUAL, out=h2)


def five_up():
    for l in range(L):
        ops.norm_linear(up_w[l], h, norm_w, None, 1e-5, False, None, ops.EPI_SILU_MUL, out=act)


def three_mlp():
    for l in range(L):
        ops.draft_mlp(h, slabs, norm_w, 1e-5, up_w[l], down_km[l], acc, F, h_out=h2)


print(f"rows {M}: per layer, us")
for name, fn in (("attention (partials)", five_attn), ("attention + merging o-proj, two launches", five_attn_oproj),
                 ("sd_draft_attn_oproj", three_attn_oproj), ("gate/up", five_up),
                 ("gate/up + down, two launches", five_mlp), ("sd_draft_mlp", three_mlp)):
    print(f"  {name:36s} {graph_time(fn):7.2f}", flush=True)
