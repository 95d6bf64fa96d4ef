"""Where a chunk of the wide linear kernel goes (diagnostic build with in-kernel clocks):
    make stamps && SD_AMD_LIB=build/stamps/libsd_amd_stamps.so python experiments/wide_stamps.py
Per case: launch time (events, 8 rotating weight copies) and, for every wave of workgroup 0, the
core clock and the time per 128-column chunk spent waiting for the chunk's weights, issuing the
next chunk's loads, in LDS reads + MFMAs, waiting for the next chunk's activations, storing them
to LDS, and in the barrier (the eight-wave kernel); for the four-wave kernel, whose loads and
stores sit inside the MFMA stream: per chunk the stream and the barrier, per launch the time
before and after the chunk loop.
    ... wide_stamps.py --quick 0|-1     192-row gate/up and LM head, eight-wave kernel | library's choice
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_omni_draft_amd import _lib, ops  # noqa: E402

lib = _lib.load()
lib.sd_debug_read_wide_stamps.argtypes = [C.c_void_p]
dev = "cuda"


def case(name, N, K, M, kind, L=8, four=False):
    gen = torch.Generator(device=dev).manual_seed(0)
    rows = 2 * N if kind == "gated" else N
    mode = ops.TILE_GATED if kind == "gated" else ops.TILE_PLAIN
    ws = [ops.tile_weights((torch.randn(rows, K, generator=gen, device=dev) * 0.02).to(torch.bfloat16), mode, 0)
          for _ in range(L)]
    x = torch.randn(M, K, generator=gen, device=dev).to(torch.bfloat16)
    res = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    epi = ops.EPI_SILU_MUL if kind == "gated" else (ops.EPI_RESIDUAL if kind == "res" else ops.EPI_STORE)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    wsb = torch.empty(int(lib.sd_linear_wide_workspace_bytes(M, N, K, epi)) + 16, dtype=torch.uint8, device=dev)
    r = res if kind == "res" else None

    def run(w):
        ops.linear_wide(w, x, None, r, epi, out=out, workspace=wsb)
    for w in ws:
        run(w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        for w in ws:
            run(w)
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (3 * L)
    st = (C.c_ulonglong * 64)()
    assert lib.sd_debug_read_wide_stamps(st) == 0
    n = max(K // 128, 1) if N * (2 if kind == "gated" else 1) >= 16384 else 1
    names = ["wait weights", "lds+mfma", "wait x", "store x", "barrier", "issue loads"]
    print(f"{name:10s} N={N:6d} K={K:6d} M={M:3d}: {us:6.1f} us/launch; {n} chunks", flush=True)
    for w in range(8):
        r = st[8 * w:8 * w + 8]
        if not r[6]:
            continue
        if four:   # (the kernel stores the two per-launch times multiplied by the chunk count)
            per = (f"stream={r[1] * 10 / n:.0f}  barrier={r[4] * 10 / n:.0f} (ns); per launch: before the loop "
                   f"{r[0] * 10 / n / 1e3:.1f} us, after it {r[2] * 10 / n / 1e3:.1f} us")
            print(f"    wave {w}: clock {r[7] / (r[6] * 10):.2f} GHz, loop {r[6] * 10 / n:.0f} ns/chunk: {per}", flush=True)
            continue
        per = "  ".join(f"{nm}={r[i] * 10 / n:.0f}" for i, nm in enumerate(names))
        print(f"    wave {w}: clock {r[7] / (r[6] * 10):.2f} GHz, loop {r[6] * 10 / n:.0f} ns/chunk: {per} (ns)", flush=True)


if __name__ == "__main__":
    if "--quick" in sys.argv:
        four = int(sys.argv[sys.argv.index("--quick") + 1])
        lib.sd_debug_wide_four_waves(four)
        case("gate/up", 14336, 4096, 192, "gated", four=four != 0)   # (1792 tiles: the four-wave kernel's case)
        case("head", 128256, 4096, 192, "plain", L=2)
        sys.exit(0)
    for M in (32, 96, 192, 256):
        case("gate/up", 14336, 4096, M, "gated")
    case("qkv-ish", 6144, 4096, 192, "plain")
    case("o", 4096, 4096, 192, "res")
    case("down", 4096, 14336, 192, "res")
    case("lm head", 128256, 4096, 192, "plain", L=2)
