"""The linear layers of a Llama-3.2-1B decoder layer at 1..2 token rows, 16 layers of distinct
weights in one replayed graph: four launches per layer (output projection, norm + gate/up, down,
next norm + QKV + rotary + cache append) against the two fused launches (sd_chain_o_up,
sd_chain_down_qkv).  Attention is left out of both (its output is a fixed buffer), so the
difference is what the fusion buys per layer.  Usage: python experiments/bench_chain.py [rows]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_omni_draft_amd import ops  # noqa: E402
from vllm_omni_draft_amd.config import create_model_config, rope_table  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 1
FP8 = len(sys.argv) > 2 and sys.argv[2] == "fp8"
L, Hq, Hkv, D, H, F = 16, 32, 8, 64, 2048, 8192
gen = torch.Generator(device="cuda").manual_seed(0)


def w(rows, K, mode, hd=0):
    W = (torch.randn(rows, K, generator=gen, device="cuda") * K ** -0.5).to(torch.bfloat16)
    if FP8:
        q, s = ops.quantize_fp8(W)
        return ops.tile_weights(q, mode, hd, scale=s)
    return ops.tile_weights(W, mode, hd)


layers = [dict(o=w(H, Hq * D, ops.TILE_PLAIN), up=w(2 * F, H, ops.TILE_GATED),
               down=w(H, F, ops.TILE_PLAIN), qkv=w((Hq + 2 * Hkv) * D, H, ops.TILE_ROTARY, D),
               g1=torch.ones(H, dtype=torch.bfloat16, device="cuda"),
               g2=torch.ones(H, dtype=torch.bfloat16, device="cuda")) for _ in range(L)]
cfg = create_model_config(None, head_dim=D, n_heads=Hq, n_kv_heads=Hkv, max_pos=2048)
cs = rope_table(cfg).cuda()
bt = torch.arange(80, dtype=torch.int32, device="cuda").reshape(1, 80)
seq_lens = torch.tensor([1100], dtype=torch.int32, device="cuda")
kc = torch.zeros(80, Hkv, D * 16, dtype=torch.bfloat16, device="cuda")
vc = torch.zeros_like(kc)
attn = (0.1 * torch.randn(M, Hq * D, generator=gen, device="cuda")).to(torch.bfloat16)
h0 = (0.1 * torch.randn(M, H, generator=gen, device="cuda")).to(torch.bfloat16)
state = ops.chain_state(H)
bytes_layer = sum(t.nbytes() for t in (layers[0]["o"], layers[0]["up"], layers[0]["down"], layers[0]["qkv"]))


def launches(h):
    q = None
    for l in layers:
        ops.linear(l["o"], attn, None, h, ops.EPI_RESIDUAL, out=h)
        act = ops.norm_linear(l["up"], h, l["g1"], None, 1e-5, False, None, ops.EPI_SILU_MUL)
        ops.linear(l["down"], act, None, h, ops.EPI_RESIDUAL, out=h)
        q = ops.norm_qkv_rope(l["qkv"], h, l["g2"], None, 1e-5, False, None, cs, kc, vc, bt, seq_lens,
                              0, 1, M, Hq, Hkv, D)
    return q


def fused(h):
    q = None
    for l in layers:
        act = ops.chain_o_up(state, l["o"], h, l["g1"], 1e-5, l["up"], 1, M, Hq, Hkv, D, attn=attn)
        q = ops.chain_down_qkv(state, l["down"], act, h, l["g2"], 1e-5, l["qkv"], cs, kc, vc, bt,
                               seq_lens, 0, 1, M, Hq, Hkv, D)
    return q


def timed(fn, name):
    h = h0.clone()
    fn(h)                       # warm-up, eager
    torch.cuda.synchronize()
    h.copy_(h0)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        q = fn(h)
    best = 1e9
    for _ in range(8):
        h.copy_(h0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    us = best * 1e3 / L
    print(f"{name}: {us:.2f} us per layer, {bytes_layer / us / 1e6:.2f} TB/s ({bytes_layer / 1e6:.1f} MB per layer)",
          flush=True)
    return h.clone(), q.clone()


ha, qa = timed(launches, "four launches")
hb, qb = timed(fused, "two fused launches")
print("rows", M, "fp8" if FP8 else "bf16", "| bit-identical:", bool(torch.equal(ha, hb) and torch.equal(qa, qb)),
      "| chain errors:", ops.chain_errors(state))
