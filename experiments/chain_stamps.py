"""In-kernel timestamps of the fused launches (workgroup 0: first streaming wave and first row
wave), from the instrumented build: `make stamps`, then
SD_AMD_LIB=build/stamps/libsd_amd_stamps.so python experiments/chain_stamps.py [rows]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_omni_draft_amd import _lib, ops  # noqa: E402
from vllm_omni_draft_amd.config import create_model_config, rope_table  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 1
Hq, Hkv, D, H, F = 32, 8, 64, 2048, 8192
gen = torch.Generator(device="cuda").manual_seed(0)
w = lambda rows, K, mode, hd=0: ops.tile_weights(
    (torch.randn(rows, K, generator=gen, device="cuda") * K ** -0.5).to(torch.bfloat16), mode, hd)
sets = [dict(o=w(H, Hq * D, ops.TILE_PLAIN), up=w(2 * F, H, ops.TILE_GATED), down=w(H, F, ops.TILE_PLAIN),
             qkv=w((Hq + 2 * Hkv) * D, H, ops.TILE_ROTARY, D)) for _ in range(6)]
g = torch.ones(H, dtype=torch.bfloat16, device="cuda")
cs = rope_table(create_model_config(None, head_dim=D, n_heads=Hq, n_kv_heads=Hkv, max_pos=2048)).cuda()
bt = torch.arange(80, dtype=torch.int32, device="cuda").reshape(1, 80)
seq_lens = torch.tensor([1100], dtype=torch.int32, device="cuda")
kc = torch.zeros(80, Hkv, D * 16, dtype=torch.bfloat16, device="cuda")
vc = torch.zeros_like(kc)
attn = (0.1 * torch.randn(M, Hq * D, generator=gen, device="cuda")).to(torch.bfloat16)
h = (0.1 * torch.randn(M, H, generator=gen, device="cuda")).to(torch.bfloat16)
state = ops.chain_state(H)
lib = _lib.load()
lib.sd_debug_read_chain_stamps.argtypes = [C.c_void_p]
buf = (C.c_uint64 * 32)()


lib.sd_debug_read_chain_wg_stamps.argtypes = [C.c_void_p]
wbuf = (C.c_uint64 * (512 * 4))()


def show_wgs(name, n_wg=256):
    """Distribution over the workgroups: when each published phase 0, had gathered phase 1's
    input and ended, relative to the earliest start."""
    assert lib.sd_debug_read_chain_wg_stamps(wbuf) == 0
    t = [[wbuf[w * 4 + i] for i in range(4)] for w in range(n_wg)]
    t0 = min(r[0] for r in t)
    cols = ["start", "published", "gathered", "end"]
    parts = []
    for i, c in enumerate(cols):
        v = sorted((r[i] - t0) / 100.0 for r in t)
        parts.append(f"{c} {v[0]:.2f}/{v[len(v) // 2]:.2f}/{v[-1]:.2f}")
    print(f"{name} all workgroups (min/median/max us): " + "  ".join(parts))


def show(name):
    torch.cuda.synchronize()
    assert lib.sd_debug_read_chain_stamps(buf) == 0
    t = list(buf)
    s = [(t[i] - t[0]) / 100.0 for i in range(5)]
    r = [(t[16 + i] - t[0]) / 100.0 for i in range(1, 5)]
    print(f"{name}: streaming wave  first barrier {s[1]:.2f}  phase 0 done {s[2]:.2f}  phase 1 input {s[3]:.2f}  "
          f"end {s[4]:.2f} us | row wave  first barrier {r[0]:.2f}  phase 0 units {r[1]:.2f}  gathered {r[2]:.2f}  "
          f"normed {r[3]:.2f} us")


for rep in range(3):
    for l in sets:      # distinct weights per call: cold caches, as in a model pass
        act = ops.chain_o_up(state, l["o"], h, g, 1e-5, l["up"], 1, M, Hq, Hkv, D, attn=attn)
        if rep == 2:
            show("o_up    ")
            show_wgs("o_up    ")
        q = ops.chain_down_qkv(state, l["down"], act, h, g, 1e-5, l["qkv"], cs, kc, vc, bt, seq_lens, 0, 1,
                               M, Hq, Hkv, D)
        if rep == 2:
            show("down_qkv")
            show_wgs("down_qkv")
print("chain errors:", ops.chain_errors(state))
