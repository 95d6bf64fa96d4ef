"""In-kernel phase times of workgroup 0 of the linear kernel (needs -DSD_LIN_STAMPS):
    make clean && make EXTRA=-DSD_LIN_STAMPS && python experiments/lin_stamps.py
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_omni_draft_amd import _lib, ops  # noqa: E402

lib = _lib.load()
lib.sd_debug_read_lin_stamps.argtypes = [C.c_void_p]
dev = "cuda"


def case(name, N, K, M, kind, warm_x=False):
    gen = torch.Generator(device=dev).manual_seed(0)
    rows = 2 * N if kind == "gated" else N
    mode = {"gated": ops.TILE_GATED}.get(kind, ops.TILE_PLAIN)
    w = ops.tile_weights((torch.randn(rows, K, generator=gen, device=dev) * 0.02).to(torch.bfloat16), mode, 0)
    x = torch.randn(M, K, generator=gen, device=dev).to(torch.bfloat16)
    x_src = x.clone()
    g = torch.ones(K, dtype=torch.bfloat16, device=dev)
    res = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    rows_t = []
    for it in range(6):
        flush.fill_(it)
        torch.cuda.synchronize()
        if warm_x:  # as in a pass: the launch before wrote the rows, everything else is cold
            x.copy_(x_src)
        if kind == "gated":
            ops.norm_linear(w, x, g, None, 1e-5, False, None, ops.EPI_SILU_MUL, out=out)
        elif kind == "norm":
            ops.norm_linear(w, x, g, None, 1e-5, False, None, ops.EPI_STORE, out=out)
        else:
            ops.linear(w, x, None, res, ops.EPI_RESIDUAL, out=out)
        torch.cuda.synchronize()
        st = (C.c_ulonglong * 16)()
        assert lib.sd_debug_read_lin_stamps(st) == 0
        rows_t.append([(st[i + 1] - st[i]) * 10 for i in range(4)])
    rows_t.sort(key=sum)
    med = rows_t[len(rows_t) // 2]
    names = ["setup+issue", "stage x", "stream+mfma", "reduce+epilogue"]
    name = name + (" warm-x" if warm_x else "")
    print(f"{name:16s} N={N} K={K} M={M}: total {sum(med)} ns  " +
          "  ".join(f"{n}={v}" for n, v in zip(names, med)), flush=True)


if __name__ == "__main__":
    case("qkv-ish", 3072, 2048, 1, "norm")
    case("gate/up", 8192, 2048, 1, "gated")
    case("o", 2048, 2048, 1, "res")
    case("down", 2048, 8192, 1, "res")
    case("qkv-ish T", 6144, 4096, 6, "norm")
    case("gate/up T", 14336, 4096, 6, "gated")
    case("down T", 4096, 14336, 6, "res")
    for args in (("qkv-ish", 3072, 2048, 1, "norm"), ("gate/up", 8192, 2048, 1, "gated"),
                 ("down", 2048, 8192, 1, "res"), ("qkv-ish T", 6144, 4096, 6, "norm"),
                 ("gate/up T", 14336, 4096, 6, "gated"), ("down T", 4096, 14336, 6, "res")):
        case(*args, warm_x=True)
