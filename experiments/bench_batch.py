"""Batched speculative decoding (the wide kernel family): Qwen2.5-7B target + Qwen2.5-0.5B draft,
k = 5, B sequences, synthetic weights and teacher-forced acceptance, next to the batched
autoregressive baseline on the same kernels.
    python experiments/bench_batch.py [B] [prompt_len] [draft_rows: auto | same | narrow | wide] [target] [draft] [fp8]
Environment: K = proposals per step (default 5); IMAGE = placeholder tokens of one synthetic image
at the start of every prompt (a target with rotary sections, e.g. qwen2-vl-7b: image rows are
random, positions come from multimodal.prepare_prompt); RAGGED=1: prompt lengths uniform in
64..prompt_len (seed 7) instead of all equal; DRAFT_SPLIT = keys per attention workgroup of the
draft's passes (default: the target's), DRAFT_LANES = parallel streams of its sub-batches; WIDE_FOUR=0|1 with a library built with
-DSD_WIDE_SWITCH (SD_AMD_LIB): the many-row linear kernel never / wherever built on four waves.
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_omni_draft_amd import config, weights  # noqa: E402
from vllm_omni_draft_amd.engine import EngineConfig, SpecEngine  # noqa: E402

if os.environ.get("WIDE_FOUR"):
    from vllm_omni_draft_amd import _lib  # noqa: E402
    _lib.load().sd_debug_wide_four_waves(int(os.environ["WIDE_FOUR"]))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
P = int(sys.argv[2]) if len(sys.argv) > 2 else 256
DRAFT_ROWS = sys.argv[3] if len(sys.argv) > 3 else "auto"
TARGET = sys.argv[4] if len(sys.argv) > 4 else "qwen2.5-7b"
DRAFT = sys.argv[5] if len(sys.argv) > 5 else "qwen2.5-0.5b"
FP8 = len(sys.argv) > 6 and sys.argv[6] == "fp8"
k, steps, warm, alpha = int(os.environ.get("K", "5")), 24, 4, 0.8
IMAGE = int(os.environ.get("IMAGE", "0"))
tcfg = config.preset(TARGET)
dcfg = config.preset(DRAFT, vocab=tcfg.vocab)  # draft vocabulary padded to the target's
tw = weights.random_init(tcfg, seed=0, fp8=FP8)
dw = weights.random_init(dcfg, seed=1, fp8=FP8)
V = min(tcfg.vocab, dcfg.vocab)
g = torch.Generator().manual_seed(7)
lens = ([int(n) for n in torch.randint(64, P + 1, (B,), generator=torch.Generator().manual_seed(7))]
        if os.environ.get("RAGGED") else [P] * B)
prompts = [torch.randint(0, V, (n,), generator=g).tolist() for n in lens]
extras = [{} for _ in range(B)]
if IMAGE:
    from vllm_omni_draft_amd import multimodal  # noqa: E402
    side = int(IMAGE ** 0.5)
    assert side * side == IMAGE and tcfg.mrope_section, "IMAGE must be a square; target needs mrope"
    grid = (1, 2 * side, 2 * side)
    V = min(V, multimodal.QWEN2_VL_IMAGE_TOKEN_ID)
    for b in range(B):
        prompts[b] = [t % V for t in prompts[b]]
        prompts[b][:IMAGE] = [multimodal.QWEN2_VL_IMAGE_TOKEN_ID] * IMAGE
        rows_ = (torch.randn(IMAGE, tcfg.hidden, generator=g) * 0.02).to(torch.bfloat16).cuda()
        emb, pos = multimodal.prepare_prompt(prompts[b], tw, rows_, [grid])
        extras[b] = dict(input_embeds=emb, rope_positions=pos)
n_new = (steps + warm) * (k + 1) + 8
max_seq = P + n_new + 16
rows = "wide" if B * (k + 1) > 16 else "narrow"


def run(eng, n_steps, n_warm):
    outs = [[eng.add_request(i, p, **extras[i])] for i, p in enumerate(prompts)]
    for _ in range(n_warm):
        for o, new in zip(outs, eng.step()):
            o.extend(new)
    torch.cuda.synchronize()
    n0 = sum(len(o) for o in outs)
    t0 = time.perf_counter()
    for _ in range(n_steps):
        for o, new in zip(outs, eng.step()):
            o.extend(new)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return outs, sum(len(o) for o in outs) - n0, dt


ar = SpecEngine(tw, None, EngineConfig(k=0, max_batch=B, max_seq_len=max_seq, rows=rows,
                                       gen_capacity=n_new + 64))
ar_out, ar_n, ar_dt = run(ar, n_new - 8 - warm, warm)
print(f"AR   B={B}: {ar_dt / (n_new - 8 - warm) * 1e3:.3f} ms/step, {ar_n / ar_dt:.1f} tok/s", flush=True)
truth = torch.zeros(B, max_seq, dtype=torch.int32)
for b in range(B):
    n = len(prompts[b])
    truth[b, :n] = torch.tensor(prompts[b])
    truth[b, n:n + len(ar_out[b])] = torch.tensor(ar_out[b])
del ar
torch.cuda.empty_cache()
eng = SpecEngine(tw, dw, EngineConfig(k=k, max_batch=B, max_seq_len=max_seq, rows=rows,
                                      draft_rows=DRAFT_ROWS, gen_capacity=n_new + 64,
                                      draft_split_tokens=int(os.environ.get("DRAFT_SPLIT", "0")),
                                      draft_lanes=int(os.environ.get("DRAFT_LANES", "4"))))
eng.set_synthetic_acceptance(truth.cuda(), alpha, seed=3)
out, n, dt = run(eng, steps, warm)
exact = all(o[:min(len(o), len(a))] == a[:min(len(o), len(a))] for o, a in zip(out, ar_out))
m = eng.metrics()
print(f"spec B={B}: {dt / steps * 1e3:.3f} ms/step, {n / dt:.1f} tok/s, "
      f"{n / steps / B:.2f} tokens/step/seq, speedup {n / dt / (ar_n / ar_dt):.3f}, "
      f"spec == AR: {exact}, family {rows}", flush=True)
