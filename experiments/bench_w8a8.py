"""fp8 activations on the wide family (W8A8) against bf16 activations on the same fp8 weights and
against bf16 weights: the 192-row gate/up launch and the other layer matrices of Llama-3-8B and
Qwen2.5-7B shapes, and a 1024-token prefill of an fp8 model with and without the option.
    python experiments/bench_w8a8.py
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_omni_draft_amd import config, ops, weights  # noqa: E402
from vllm_omni_draft_amd.engine import EngineConfig, SpecEngine  # noqa: E402

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


def case(name, N, K, M, kind, L=8):
    gen = torch.Generator(device=dev).manual_seed(0)
    rows = 2 * N if kind == "gated" else N
    mode = ops.TILE_GATED if kind == "gated" else ops.TILE_PLAIN
    bf, f8 = [], []
    for _ in range(L):
        W = (torch.randn(rows, K, generator=gen, device=dev) * 0.02).to(torch.bfloat16)
        bf.append(ops.tile_weights(W, mode, 0))
        q, s = ops.quantize_fp8(W)
        f8.append(ops.tile_weights(q, mode, 0, scale=s))
        del W, q
    x = torch.randn(M, K, generator=gen, device=dev).to(torch.bfloat16)
    res = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    epi = ops.EPI_SILU_MUL if kind == "gated" else (ops.EPI_RESIDUAL if kind == "res" else ops.EPI_STORE)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    wsb = torch.empty(int(ops._lib.load().sd_linear_wide_workspace_bytes(M, N, K, epi)) + 16,
                      dtype=torch.uint8, device=dev)
    r = res if kind == "res" else None
    t_bf = timed(lambda w: ops.linear_wide(w, x, None, r, epi, out=out, workspace=wsb), bf)
    t_w8 = timed(lambda w: ops.linear_wide(w, x, None, r, epi, out=out, workspace=wsb), f8)
    t_a8 = timed(lambda w: ops.linear_wide(w, x, None, r, epi, out=out, workspace=wsb, fp8_activations=True), f8)
    flops = 2.0 * M * rows * K
    print(f"{name:16s} N={N:6d} K={K:6d} M={M:3d}: bf16 {t_bf:7.1f} us | fp8 weights {t_w8:7.1f} us | "
          f"fp8 weights + activations {t_a8:7.1f} us (row scales included; {flops / t_a8 * 1e-6:6.0f} TFLOP/s)",
          flush=True)


def prefill(name, n_prompt, fp8_activations):
    cfg = config.preset(name, n_layers=8)
    w = weights.random_init(cfg, seed=0, device=dev, fp8=True)
    eng = SpecEngine(w, None, EngineConfig(k=0, max_batch=1, max_seq_len=n_prompt + 64,
                                           fp8_activations=fp8_activations))
    prompt = torch.randint(0, cfg.vocab, (n_prompt,), generator=torch.Generator().manual_seed(1)).tolist()
    ts = []
    for _ in range(4):
        eng.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.add_request(0, prompt)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3


if __name__ == "__main__":
    for M in (128, 192, 256):
        print(f"--- Llama-3-8B shapes, M={M}")
        case("qkv", 6144, 4096, M, "plain")
        case("o (+res)", 4096, 4096, M, "res")
        case("gate/up (silu)", 14336, 4096, M, "gated")
        case("down (+res)", 4096, 14336, M, "res")
    print("--- Qwen2.5-7B shapes, M=192")
    case("gate/up (silu)", 18944, 3584, 192, "gated")
    case("down (+res)", 3584, 18944, 192, "res")
    for name in ("llama-3-8b", "qwen2.5-7b"):
        a, b = prefill(name, 1024, False), prefill(name, 1024, True)
        print(f"prefill of 1024 tokens, {name} shapes, 8 layers + LM head, fp8 weights: {a:.2f} ms with bf16 "
              f"activations, {b:.2f} ms with fp8 activations")
