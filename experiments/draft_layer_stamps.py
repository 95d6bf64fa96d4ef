"""In-kernel phase times of the LAST workgroup of sd_draft_attn_oproj and sd_draft_mlp (a build
with -DSD_DRAFT_STAMPS: `make stamps`, loaded through SD_AMD_LIB), Llama-3.2-1B widths, one row:
    make stamps && SD_AMD_LIB=build/stamps/libsd_amd_stamps.so python experiments/draft_layer_stamps.py
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_omni_draft_amd import _lib, ops  # noqa: E402

lib = _lib.load()
lib.sd_debug_read_draft_stamps.argtypes = [C.c_void_p]
dev = "cuda"
H, F, Hq, Hkv, D, ctx, max_ctx = 2048, 8192, 32, 8, 64, 1100, 2048
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1
q_len = 1 if M == 1 else 2
B = M // q_len


def rnd(*shape, s=0.03):
    return (torch.randn(*shape, device=dev) * s).to(torch.bfloat16)


pages = (max_ctx + 15) // 16
kc, vc = rnd(B * pages, Hkv, D * 16, s=0.5), rnd(B * pages, Hkv, D * 16, s=0.5)
bt = torch.randperm(B * pages, device=dev).to(torch.int32).view(B, pages).contiguous()
sl = torch.full((B,), ctx, dtype=torch.int32, device=dev)
q = rnd(M, Hq * D, s=1.0)
o_w = ops.tile_weights(rnd(H, Hq * D))
up_w = ops.tile_weights(rnd(2 * F, H), ops.TILE_GATED)
down_km = ops.kmajor_tiles(ops.tile_weights(rnd(H, F)))
norm_w = torch.ones(H, dtype=torch.bfloat16, device=dev)
h = rnd(M, H, s=1.0)
acc = torch.zeros(M, H, dtype=torch.float32, device=dev)
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
rows = {0: [], 1: []}
for it in range(7):
    flush.fill_(it)
    torch.cuda.synchronize()
    slabs = ops.draft_attn_oproj(q, kc, vc, bt, sl, 0, o_w, B, q_len, Hq, Hkv, D, D ** -0.5, max_ctx, zero=acc)
    torch.cuda.synchronize()
    flush.fill_(it + 1)
    torch.cuda.synchronize()
    ops.draft_mlp(h, slabs, norm_w, 1e-5, up_w, down_km, acc, F)
    torch.cuda.synchronize()
    st = (C.c_ulonglong * 32)()
    assert lib.sd_debug_read_draft_stamps(st) == 0
    rows[0].append([(st[i + 1] - st[i]) * 10 for i in range(5)])
    rows[1].append([(st[16 + i + 1] - st[16 + i]) * 10 for i in range(7)])
names = {0: ["prologue (bt, q, len)", "attention loop", "wave merge", "normalise", "projection + store"],
         1: ["staging loads + sums", "normalise", "gate/up stream", "cross-wave store", "activation", "down mfma",
             "atomics issue"]}
for k, title in ((0, "sd_draft_attn_oproj"), (1, "sd_draft_mlp")):
    r = sorted(rows[k], key=sum)[len(rows[k]) // 2]
    print(f"{title} (rows {M}, cold caches): total {sum(r)} ns  " + "  ".join(f"{n}={v}" for n, v in zip(names[k], r)),
          flush=True)
