#!/usr/bin/env python3
"""Headline benchmark: output tokens/s and accept-len of speculative decoding,
Llama-3-8B target + Llama-3.2-1B draft, k = 5, one sequence per GPU.

    python bench.py --gpus 1 --steps 64 --warmup 8
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one speculative step of the hot path (k draft passes, one
(k+1)-token target verify pass over the paged KV cache, acceptance, state
update), replayed as one hipGraph; the host reads the emitted tokens back after
every step, as a serving loop must.  Data is synthetic: seeded random-init
weights of the named architectures and a random-token prompt.  Random weights
give a random draft no chance of agreeing with the target, so acceptance is a
controlled parameter: the greedy autoregressive stream of the target is
recorded first, and each draft proposal (computed in full by the draft model)
is replaced by the true next token with probability --accept-prob.  The run
then also proves, at full size, that the speculative stream equals the
autoregressive stream token for token.

With N > 1 every rank runs an independent replica (its own sequence); no
collective is on the data path; throughput is summed (weak scaling).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E datasheet peak


PMC_FILE = "profiles/r3_pmc_hbm.json"


def pmc_traffic(kernel_substr):
    """HBM bytes per launch (read + write) of a kernel, NOT measured in this run: read from
    the committed counter summary PMC_FILE (separate FETCH_SIZE / WRITE_SIZE passes of
    rocprofv3 over experiments/pmc_*.py, FETCH_SIZE doubled as gfx950 requires), or None.
    The bench line names the file as `traffic_source`."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), *PMC_FILE.split("/"))
    try:
        with open(path) as f:
            ks = json.load(f)["kernels"]
    except (OSError, ValueError, KeyError):
        return None
    for name, e in ks.items():
        if kernel_substr in name and "hbm_read_bytes_per_launch" in e:
            return e["hbm_read_bytes_per_launch"] + e.get("hbm_write_bytes_per_launch", 0)
    return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--k", type=int, default=5)
    ap.add_argument("--prompt-len", type=int, default=1024)
    ap.add_argument("--accept-prob", type=float, default=0.8)
    ap.add_argument("--target", default="llama-3-8b")
    ap.add_argument("--draft", default="llama-3.2-1b")
    ap.add_argument("--split-tokens", type=int, default=256)
    ap.add_argument("--draft-split-tokens", type=int, default=0)
    ap.add_argument("--weights", choices=["bf16", "fp8"], default="bf16",
                    help="fp8: both models' matrices quantised at load (e4m3, per-row scales), "
                         "bf16 activations; an extra data point, the headline is bf16")
    ap.add_argument("--read-back", choices=["overlapped", "sync"], default="sync",
                    help="host read-back of a step's tokens: while the next step runs, or before it "
                         "is enqueued (both rates are reported; `value` is this one)")
    ap.add_argument("--no-sweep", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--draft-layers", choices=["three", "five"], default="five",
                    help="five: the draft's passes on the five-launch layer (the target's kernels); "
                         "three: three launches per layer where the shapes allow it (measured no faster)")
    ap.add_argument("--read-ahead-mb", default="0,0", metavar="TARGET,DRAFT",
                    help="MiB of the next launches' weights that each layer's attention launch reads "
                         "ahead on its idle compute units, for the target's and the draft's passes "
                         "(EngineConfig.target_read_ahead_mb / draft_read_ahead_mb; same bits)")
    ap.add_argument("--layers", type=int, default=0, help="debug only: truncate both models")
    ap.add_argument("--cpu-baseline-worker", action="store_true",
                    help="internal: the child process of the cpu_baseline leg (SD_AMD_BACKEND=cpu)")
    ap.add_argument("--cpu-budget", type=float, default=25.0)
    # The tensor-parallel configuration (BASELINE.json configs[3]): ONE job over all ranks.
    ap.add_argument("--tp", type=int, default=int(os.environ.get("SD_BENCH_TP", "0") or 0),
                    help="tensor-parallel degree (= WORLD_SIZE): the target's verify pass is sharded over "
                         "the ranks, one sequence for the whole job; target / draft / k default to "
                         "llama-3-70b / llama-3-8b / 4")
    ap.add_argument("--tp-collective", choices=["rccl", "ipc"], default="rccl",
                    help="rccl: the library's RCCL group; ipc: one-shot all-reduce kernels over "
                         "peer-mapped memory")
    ap.add_argument("--tp-draft", choices=["replicated", "sharded"], default="replicated",
                    help="the draft whole on every rank (no exchange in its passes), or sharded "
                         "like the target with a vocabulary-parallel head")
    return ap.parse_args()


def load_weights(name, seed, device, layers=0, fp8=False, head_subset=None):
    from vllm_omni_draft_amd import config, weights
    root = os.environ.get("SD_MODEL_ROOT", "")
    path = os.path.join(root, name) if root else ""
    if path and os.path.isdir(path):
        cfg = config.load_hf_config(path)
        return (weights.from_safetensors_dir(cfg, path, device, fp8=fp8, head_subset=head_subset),
                "local safetensors")
    cfg = config.preset(name, **({"n_layers": layers} if layers else {}))
    return (weights.random_init(cfg, seed=seed, device=device, fp8=fp8, head_subset=head_subset),
            "seeded random init")


def streamed_weight_bytes(w):
    """Bytes of weight matrices (and fp8 scales) one forward pass of the model streams."""
    n = w.lm_head.nbytes()
    for l in w.layers:
        n += l.qkv_w.nbytes() + l.o_w.nbytes() + l.up_w.nbytes() + l.down_w.nbytes()
    return n


def run_generation(eng, prompt, n_steps, warmup, dist_barrier, overlapped=False):
    """Prefill, `warmup` untimed steps, then exactly n_steps timed steps, every step's tokens
    read back by the host inside the timed region: right after the step (synchronous), or
    while the next step runs (overlapped: step t+1 is enqueued before step t is read, the
    last step is flushed before the clock stops).  Returns (tokens, timed_tokens, seconds,
    prefill_s)."""
    def steps(n):
        if not overlapped:
            for _ in range(n):
                out.extend(eng.step()[0])
            return
        for _ in range(n):
            r = eng.step_overlapped()
            if r:
                out.extend(r[0])
        out.extend(eng.flush()[0])

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = [eng.add_request(0, prompt)]
    torch.cuda.synchronize()
    prefill_s = time.perf_counter() - t0
    steps(warmup)
    dist_barrier()
    torch.cuda.synchronize()
    n0 = len(out)
    t0 = time.perf_counter()
    steps(n_steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dist_barrier()
    return out, len(out) - n0, dt, prefill_s


def step_bytes(tcfg, dcfg, k, ctx, t_weight_bytes=None, d_weight_bytes=None):
    """Algorithmic HBM bytes of one speculative step at context length ctx (weight bytes
    default to the bf16 sizes of the configurations)."""
    V = tcfg.vocab
    kv_t = tcfg.kv_bytes_per_token() * ctx
    kv_d = dcfg.kv_bytes_per_token() * ctx * k
    logits = (k + 1) * V * 4 * 2 + k * V * 4 * 2  # written once, read once by argmax
    wt = tcfg.param_bytes() if t_weight_bytes is None else t_weight_bytes
    wd = dcfg.param_bytes() if d_weight_bytes is None else d_weight_bytes
    return wt + k * wd + kv_t + kv_d + logits


def kernel_roofline(eng, tw, rows, reps=3):
    """Times the dominant kernel of the step — the gate/up weight-streaming
    linear of the target verify pass (M = k+1 rows) — with events on the
    engine's stream, cycling through every layer's matrix so that no launch
    finds its weights in the 256 MiB Infinity Cache."""
    from vllm_omni_draft_amd import ops
    cfg = tw.cfg
    x = torch.randn(rows, cfg.hidden, device=eng.device).to(torch.bfloat16)
    out = torch.empty(rows, cfg.ffn, dtype=torch.bfloat16, device=eng.device)
    epi = ops.EPI_SILU_MUL if cfg.arch == "llama" else ops.EPI_RELU
    torch.cuda.synchronize()  # x was filled on another stream
    with torch.cuda.stream(eng.stream):
        for l in tw.layers:  # warm-up pass
            ops.norm_linear(l.up_w, x, l.mlp_norm_w, l.mlp_norm_b, cfg.norm_eps, cfg.arch != 'llama', l.up_b, epi, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            for l in tw.layers:
                ops.norm_linear(l.up_w, x, l.mlp_norm_w, l.mlp_norm_b, cfg.norm_eps, cfg.arch != 'llama', l.up_b, epi, out=out)
        e1.record()
        e1.synchronize()
    n = reps * len(tw.layers)
    us = e0.elapsed_time(e1) * 1e3 / n
    w = tw.layers[0].up_w
    bytes_per_launch = w.nbytes() + x.numel() * 2 + out.numel() * 2
    gbs = bytes_per_launch / us * 1e-3
    n_rows = 2 * w.N if cfg.arch == "llama" else w.N
    wdt = "bf16" if w.scale is None else "fp8"
    return {"kernel": f"linear_kernel<norm+gate_up+silu_mul>, M={rows} [{n_rows}x{w.K}] {wdt}",
            "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(gbs / HBM_PEAK_GBS, 4),
            "traffic": pmc_traffic("linear_kernel<2, 1, false, false>") if rows == 6 and w.scale is None else None,
            "traffic_source": PMC_FILE,
            "bytes_per_launch": bytes_per_launch, "avg_us": round(us, 2), "launches": n}


def draft_pass_roofline(dw, device, ctx, split_tokens, fused_layers=False, read_ahead_mb=0):
    """One decode pass of the draft model (one row, `ctx` cached tokens, LM head included) in a
    replayed graph of four passes on a cache of its own: the bytes a pass must stream (every
    weight matrix once, the context's K/V once) over its duration.  The k passes of a step run
    one after the other, so this is the rate the draft side of the step runs at."""
    import ctypes as C
    from vllm_omni_draft_amd import _lib
    from vllm_omni_draft_amd.kv_cache import PagedKVCache
    from vllm_omni_draft_amd.model_runner import DecoderModel
    cfg = dw.cfg
    max_ctx = ctx + 64
    model = DecoderModel(dw, max_ctx, split_tokens, fused_layers=fused_layers, read_ahead_mb=read_ahead_mb)
    kv = PagedKVCache(cfg, PagedKVCache.pages_for(cfg, 1, max_ctx), 1, max_ctx, device)
    kv.ensure_capacity(0, max_ctx)
    kv.k.normal_(0, 0.5)
    kv.v.normal_(0, 0.5)
    seq_lens = torch.tensor([ctx], dtype=torch.int32, device=device)
    toks = torch.arange(5, 13, dtype=torch.int32, device=device)
    logits = torch.empty(1, cfg.head_rows, dtype=torch.float32, device=device)
    lib = _lib.load()
    stream = torch.cuda.Stream()
    reps = 4
    enqueue = lambda: [model.forward(kv, toks, 0, 0, seq_lens, 0, 1, 1, logits, max_ctx) for _ in range(reps)]
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        enqueue()
        torch.cuda.synchronize()
        g = C.c_void_p()
        _lib.check(lib.sd_graph_begin(stream.cuda_stream), "sd_graph_begin")
        enqueue()
        _lib.check(lib.sd_graph_end(stream.cuda_stream, C.byref(g)), "sd_graph_end")
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            lib.sd_graph_launch(g, stream.cuda_stream)
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / reps)
        lib.sd_graph_destroy(g)
    us = sorted(ts)[len(ts) // 2]
    nbytes = streamed_weight_bytes(dw) + 2 * cfg.n_layers * cfg.n_kv_heads * cfg.head_dim * ctx * 2
    gbs = nbytes / us * 1e-3
    return {"what": f"one decode pass of the draft ({cfg.n_layers} layers + LM head), 1 row, {ctx} cached "
                    f"tokens, in a replayed graph", "bound": "hbm", "bytes_per_pass": int(nbytes),
            "us_per_pass": round(us, 1),
            "launches_per_pass": (3 if model.fused_layers else 5) * cfg.n_layers + 2,
            "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(gbs / HBM_PEAK_GBS, 4)}


def attention_roofline(device, k):
    """The verify attention kernel alone at a KV-bound size: 32 sequences x 4096
    cached tokens, Llama-3-8B head shape, k+1 query tokens each."""
    from vllm_omni_draft_amd import ops
    B, ctx, Hq, Hkv, D, q_len = 32, 4096, 32, 8, 128, k + 1
    pages_per_seq = (ctx + q_len + 15) // 16
    n_pages = B * pages_per_seq
    kc = (torch.randn(n_pages, Hkv, D * 16, device=device) * 0.5).to(torch.bfloat16)
    vc = (torch.randn(n_pages, Hkv, D * 16, device=device) * 0.5).to(torch.bfloat16)
    bt = torch.randperm(n_pages, device=device).to(torch.int32).view(B, pages_per_seq).contiguous()
    q = torch.randn(B * q_len, Hq * D, device=device).to(torch.bfloat16)
    sl = torch.full((B,), ctx, dtype=torch.int32, device=device)
    ws = torch.empty(ops.paged_attention_workspace_bytes(B, q_len, Hq, D, ctx + q_len, 4096),
                     dtype=torch.uint8, device=device)
    run = lambda: ops.paged_attention(q, kc, vc, bt, sl, 0, B, q_len, Hq, Hkv, D, D ** -0.5,
                                      ctx + q_len, 4096, workspace=ws)
    # Checked before it is timed: two sequences get K/V of known logical content (written into
    # their pages in the cache layout), and the launch's output rows for them are compared with
    # the CPU oracle's attention, one KV-head group each.
    checked = _check_attention(run, kc, vc, bt, q, (0, B - 1), ctx, q_len, Hq, Hkv, D)
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    kv_bytes = B * (ctx + q_len) * Hkv * D * 2 * 2
    gbs = kv_bytes / us * 1e-3
    return {"kernel": "paged_attn_kernel<D=128> + merge, B=32 ctx=4096 q_len=%d, 4096 keys per workgroup" % q_len,
            "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": pmc_traffic("paged_attn_kernel<128, 1"),
            "traffic_source": PMC_FILE, "checked": checked,
            "bytes_per_launch": kv_bytes, "avg_us": round(us, 2)}


def attention_roofline_headline(device, k, ctx, split_tokens, max_ctx):
    """The verify attention as the headline step really launches it: one sequence, ctx cached
    tokens, k+1 query rows, `split_tokens` keys per workgroup up to the context bucket max_ctx,
    the split merge launch included; a replayed graph of 32 launches over 32 distinct caches
    (Llama-3-8B head shape).  Latency-bound: a few MB of K/V over 40-64 workgroups."""
    import ctypes as C
    from vllm_omni_draft_amd import _lib, ops
    Hq, Hkv, D, q_len, L = 32, 8, 128, k + 1, 32
    pages = (max_ctx + 15) // 16
    kcs = [(torch.randn(pages, Hkv, D * 16, device=device) * 0.5).to(torch.bfloat16) for _ in range(L)]
    vcs = [(torch.randn(pages, Hkv, D * 16, device=device) * 0.5).to(torch.bfloat16) for _ in range(L)]
    bt = torch.randperm(pages, device=device).to(torch.int32).view(1, pages).contiguous()
    q = torch.randn(q_len, Hq * D, device=device).to(torch.bfloat16)
    sl = torch.full((1,), ctx, dtype=torch.int32, device=device)
    ws = torch.empty(ops.paged_attention_workspace_bytes(1, q_len, Hq, D, max_ctx, split_tokens),
                     dtype=torch.uint8, device=device)
    lib = _lib.load()
    stream = torch.cuda.Stream()

    def enqueue():
        for kc, vc in zip(kcs, vcs):
            ops.paged_attention(q, kc, vc, bt, sl, 0, 1, q_len, Hq, Hkv, D, D ** -0.5, max_ctx,
                                split_tokens, workspace=ws)
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        enqueue()
        torch.cuda.synchronize()
        g = C.c_void_p()
        _lib.check(lib.sd_graph_begin(stream.cuda_stream), "sd_graph_begin")
        enqueue()
        _lib.check(lib.sd_graph_end(stream.cuda_stream, C.byref(g)), "sd_graph_end")
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            lib.sd_graph_launch(g, stream.cuda_stream)
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / L)
        lib.sd_graph_destroy(g)
    us = sorted(ts)[len(ts) // 2]
    kv_bytes = (ctx + q_len) * Hkv * D * 2 * 2
    gbs = kv_bytes / us * 1e-3
    return {"kernel": f"paged_attn_kernel<D=128> + merge, B=1 ctx={ctx} q_len={q_len}, {split_tokens} keys per "
                      f"workgroup, context bucket {max_ctx}", "bound": "hbm (latency-bound at this size)",
            "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(gbs / HBM_PEAK_GBS, 4), "bytes_per_launch": kv_bytes, "avg_us": round(us, 2)}


def grid_point(tw, dw, B, prompt_len, k, steps, warmup, accept_prob, use_graph=True, temperature=0.0,
               split_tokens=256):
    """One more point of the measurement grid: B sequences with prompts of prompt_len random
    tokens, autoregressive baseline on the same kernel family first (its stream is the truth the
    acceptance schedule is imposed with), then the speculative run.  temperature > 0: seeded
    sampling on both sides; acceptance cannot be imposed then (it is what random weights give)
    and the streams are not comparable token for token."""
    from vllm_omni_draft_amd.engine import EngineConfig, SpecEngine
    V = min(tw.cfg.vocab, dw.cfg.vocab)
    g = torch.Generator().manual_seed(4321 + prompt_len + B)
    prompts = [torch.randint(0, V, (prompt_len,), generator=g).tolist() for _ in range(B)]
    n_new = (steps + warmup) * (k + 1) + 8
    max_seq = prompt_len + n_new + 16
    rows = "wide" if B * (k + 1) > 16 else "narrow"
    sampling = temperature > 0.0

    def run(eng, n_steps, n_warm):
        kw = [dict(seed=1000 + i) if sampling else {} for i in range(B)]
        t0 = time.perf_counter()
        outs = [[eng.add_request(i, p, **kw[i])] for i, p in enumerate(prompts)]
        torch.cuda.synchronize()
        prefill = time.perf_counter() - t0
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
        return outs, sum(len(o) for o in outs) - n0, time.perf_counter() - t0, prefill

    common = dict(max_batch=B, max_seq_len=max_seq, rows=rows, gen_capacity=n_new + 64,
                  use_graph=use_graph, temperature=temperature, split_tokens=split_tokens)
    ar = SpecEngine(tw, None, EngineConfig(k=0, **common))
    ar_steps = n_new - 8 - warmup if not sampling else steps * 2
    ar_out, ar_n, ar_dt, _ = run(ar, ar_steps, warmup)
    del ar
    torch.cuda.empty_cache()
    eng = SpecEngine(tw, dw, EngineConfig(k=k, **common))
    if not sampling:
        truth = torch.zeros(B, max_seq, dtype=torch.int32)
        for b in range(B):
            truth[b, :prompt_len] = torch.tensor(prompts[b])
            truth[b, prompt_len:prompt_len + len(ar_out[b])] = torch.tensor(ar_out[b])
        eng.set_synthetic_acceptance(truth.to(tw.embed.device), accept_prob, seed=-warmup,
                                     pattern_len=min(steps, 64))
    out, n, dt, prefill = run(eng, steps, warmup)
    exact = None if sampling else all(o[:min(len(o), len(a))] == a[:min(len(o), len(a))]
                                      for o, a in zip(out, ar_out))
    del eng
    torch.cuda.empty_cache()
    return {"batch": B, "prompt_len": prompt_len, "k": k, "temperature": temperature, "kernel_family": rows,
            "tokens_per_s": round(n / dt, 1), "ms_per_step": round(dt / steps * 1e3, 4),
            "accept_len": round(n / steps / B - 1.0, 3), "ar_tokens_per_s": round(ar_n / ar_dt, 1),
            "speedup_vs_ar": round(n / dt / (ar_n / ar_dt), 3), "spec_stream_equals_ar_stream": exact,
            "prefill_s": round(prefill, 3)}


def _check_attention(run, kc, vc, bt, q, seqs, ctx, q_len, Hq, Hkv, D):
    """Output rows of the timed attention launch against oracle.attention (the checker, on the
    CPU) for the sequences `seqs`.  Raises on a mismatch; returns True."""
    from oracle.attention import verify_attention
    dev = q.device
    n, G = ctx + q_len, Hq // Hkv
    pos = torch.arange(n, device=dev)
    slot = pos % 16
    vslot = (slot & 3) | (((slot >> 3) & 1) << 2) | (((slot >> 2) & 1) << 3)
    d = torch.arange(D, device=dev)
    k_idx = (d // 8)[None, :] * 128 + slot[:, None] * 8 + (d % 8)[None, :]
    v_idx = d[None, :] * 16 + vslot[:, None]
    h = torch.arange(Hkv, device=dev)
    logical = {}
    for b in seqs:
        k = (torch.randn(n, Hkv, D, device=dev) * 0.5).to(torch.bfloat16)
        v = (torch.randn(n, Hkv, D, device=dev) * 0.5).to(torch.bfloat16)
        page = bt[b].long()[pos // 16]
        kc[page[:, None, None], h[None, :, None], k_idx[:, None, :]] = k
        vc[page[:, None, None], h[None, :, None], v_idx[:, None, :]] = v
        logical[b] = (k.cpu(), v.cpu())
    out = run()
    torch.cuda.synchronize()
    out = out.float().cpu().view(-1, q_len, Hq, D)
    qc = q.float().cpu().view(-1, q_len, Hq, D)
    for i, b in enumerate(seqs):
        kh = (3 * i + 1) % Hkv                        # one KV-head group per checked sequence
        k, v = logical[b]
        ref = verify_attention(qc[b][:, kh * G:(kh + 1) * G], k[:, kh:kh + 1], v[:, kh:kh + 1],
                               [ctx + t for t in range(q_len)], D ** -0.5)
        torch.testing.assert_close(out[b][:, kh * G:(kh + 1) * G], ref, rtol=2e-2, atol=2e-2)
    return True


def cpu_baseline(budget_s=25.0):
    """The CPU leg: this repository's own CPU build of the C ABI (libsd_cpu.so, C++/OpenMP)
    driven by the same SpecDecodeWorker, on BASELINE.json configs[0] shapes (OPT-125m target and
    draft, k = 3, greedy, batch 8, seeded random weights), in a child process (the backend is a
    per-process choice) on this box's host cores.  The reference has no CPU speculative-decoding
    path of its own to time (its dependency's CPU backend is not installed).  Bounded: about
    budget_s seconds of steps."""
    import subprocess
    env = dict(os.environ, SD_AMD_BACKEND="cpu")
    env.pop("SD_AMD_LIB", None)
    cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16") or 16))
    env["OMP_NUM_THREADS"] = str(cores)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker",
                        "--cpu-budget", str(budget_s)], env=env, capture_output=True, text=True,
                       timeout=600)
    if r.returncode != 0:
        raise RuntimeError(f"cpu baseline worker failed: {r.stderr[-500:]}")
    out = json.loads(r.stdout.strip().splitlines()[-1])
    out["cores"] = cores
    return out


def cpu_baseline_worker(budget_s):
    """Runs inside the child process (SD_AMD_BACKEND=cpu); prints one JSON line."""
    from vllm_omni_draft_amd import _lib, config, weights
    from vllm_omni_draft_amd.spec_decode_worker import (ExecuteModelRequest, SequenceGroupMetadata,
                                                        SpecDecodeConfig, SpecDecodeWorker)
    assert _lib.backend() == "cpu"
    cfg = config.preset("opt-125m", max_pos=512)
    w = weights.random_init(cfg, seed=125, device="cpu")
    k, B, n_prompt = 3, 8, 128
    worker = SpecDecodeWorker(SpecDecodeConfig(target=w, draft=w, num_speculative_tokens=k,
                                               max_num_seqs=B, max_model_len=400, rows="narrow"))
    worker.init_device()
    worker.load_model()
    g = torch.Generator().manual_seed(1234)
    prompts = {f"r{i}": torch.randint(0, cfg.vocab, (n_prompt,), generator=g).tolist() for i in range(B)}
    t0 = time.perf_counter()
    worker.execute_model(ExecuteModelRequest([SequenceGroupMetadata(r, True, p) for r, p in prompts.items()]))
    prefill_s = time.perf_counter() - t0
    running = ExecuteModelRequest([SequenceGroupMetadata(r, False) for r in prompts])
    worker.execute_model(running)            # warm-up step
    steps = tokens = 0
    t0 = time.perf_counter()
    while steps < 3 or (time.perf_counter() - t0 < budget_s and steps < 60):
        out = worker.execute_model(running)
        tokens += sum(len(t) for t in out.sampled_token_ids)
        steps += 1
    dt = time.perf_counter() - t0
    print(json.dumps({
        "value": round(tokens / dt, 2), "unit": "tokens/s", "cores": None,
        "kind": "port (own C-ABI CPU backend: libsd_cpu.so, C++/OpenMP, behind the same SpecDecodeWorker)",
        "sample": f"{steps} speculative steps of BASELINE.json configs[0] shapes (opt-125m target + "
                  f"opt-125m draft with the same seeded random weights, k={k}, greedy, batch {B}, "
                  f"prompt {n_prompt}): {dt / steps:.3f} s/step, {tokens / steps / B:.2f} tokens per step "
                  f"and sequence (every proposal accepted: draft == target), prefill {prefill_s:.1f} s"}))


def main_tp(args):
    """One tensor-parallel job over WORLD_SIZE ranks (one process per GPU): the target's shards
    with seeded random values (every rank draws its own; norms and embedding are the same on all
    ranks), the draft replicated or sharded, every rank running the same steps in lockstep; the
    verify pass's two all-reduces per layer and the head's slice-maxima exchange go through the
    chosen collective inside the step graph.  value = the job's tokens/s (one sequence)."""
    from vllm_omni_draft_amd import config, parallel, weights
    from vllm_omni_draft_amd.engine import EngineConfig, SpecEngine
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("SD_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    if world != args.tp:
        raise SystemExit(f"--tp {args.tp} needs WORLD_SIZE == {args.tp} (launch with torch.distributed.run)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU path)")
    backend = os.environ.get("SD_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    red_device = device if backend == "nccl" else torch.device("cpu")
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world > 1 or not dist.is_initialized():
        dist.init_process_group(backend, **({"device_id": device} if backend == "nccl" else {}))
    target = args.target if args.target != "llama-3-8b" else "llama-3-70b"
    draft = args.draft if args.draft != "llama-3.2-1b" else "llama-3-8b"
    k = args.k if args.k != 5 else 4
    over = {"n_layers": args.layers} if args.layers else {}
    full_t, full_d = config.preset(target, **over), config.preset(draft, **over)
    tcfg = full_t.tp_shard(world, vocab_parallel=True)
    sharded_draft = args.tp_draft == "sharded"
    dcfg = full_d.tp_shard(world, vocab_parallel=True) if sharded_draft else full_d
    if args.tp_collective == "ipc":
        group = parallel.IpcTensorParallelGroup(rank, world, max_message_bytes=256 * full_t.hidden * 2)
    else:
        group = parallel.TensorParallelGroup(rank, world)
    tw = weights.random_init(tcfg, seed=1, device=device, tp_rank=rank)
    dw = weights.random_init(dcfg, seed=2, device=device, tp_rank=rank if sharded_draft else 0)
    torch.cuda.empty_cache()     # the f32 temporaries of the initialisation go back to the device
    free_b, total_b = torch.cuda.mem_get_info()
    print(f"[rank {rank}] weights {(streamed_weight_bytes(tw) + streamed_weight_bytes(dw)) / 2**30:.1f} GiB, "
          f"reserved {torch.cuda.memory_reserved() / 2**30:.1f} GiB, device free {free_b / 2**30:.1f} of "
          f"{total_b / 2**30:.1f} GiB", file=sys.stderr, flush=True)
    g = torch.Generator().manual_seed(1234)
    prompt = torch.randint(0, full_t.vocab, (args.prompt_len,), generator=g).tolist()
    n_truth = (args.warmup + args.steps) * (k + 1) + 8
    max_seq = args.prompt_len + n_truth + 16

    def engine(kk, d=None):
        return SpecEngine(tw, d, EngineConfig(k=kk, max_seq_len=max_seq, split_tokens=args.split_tokens,
                                              draft_split_tokens=args.draft_split_tokens,
                                              use_graph=not args.no_graph, gen_capacity=n_truth + 64),
                          tp_group=group, draft_tp=sharded_draft and d is not None)

    ar = engine(0)
    torch.cuda.synchronize()
    dist.barrier()       # the ranks' kernels wait for each other with bounded waits: start together
    warm = 0
    if hasattr(group, "warm_up"):
        t0 = time.perf_counter()
        warm = group.warm_up(256 * full_t.hidden * 2)
        print(f"[rank {rank}] first exchange: {time.perf_counter() - t0:.2f} s, {warm} waits gave up",
              file=sys.stderr, flush=True)
        dist.barrier()
    if hasattr(group, "timeouts"):
        ar._tp_timeouts_seen = group.timeouts()
    ar_tokens, ar_n, ar_dt, _ = run_generation(ar, prompt, n_truth - 4 - args.warmup, args.warmup, dist.barrier)
    truth = torch.zeros(1, max_seq, dtype=torch.int32)
    truth[0, :len(prompt)] = torch.tensor(prompt)
    truth[0, len(prompt):len(prompt) + len(ar_tokens)] = torch.tensor(ar_tokens)
    del ar
    torch.cuda.empty_cache()
    eng = engine(k, dw)
    if hasattr(group, "timeouts"):
        eng._tp_timeouts_seen = group.timeouts()
    eng.set_synthetic_acceptance(truth.to(device), args.accept_prob, seed=-args.warmup,
                                 pattern_len=min(args.steps, 64))
    torch.cuda.synchronize()
    dist.barrier()
    toks, n_timed, dt, prefill = run_generation(eng, prompt, args.steps, args.warmup, dist.barrier)
    n_cmp = min(len(toks), len(ar_tokens))
    exact = toks[:n_cmp] == ar_tokens[:n_cmp]
    t = torch.tensor([dt, ar_dt], dtype=torch.float64, device=red_device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    # every rank emitted the same stream: a checksum of it must agree, and all must be exact
    chk = float(sum((i + 1) * tok for i, tok in enumerate(toks)) % 1000003)
    c = torch.tensor([chk, -chk, float(exact)], dtype=torch.float64, device=red_device)
    dist.all_reduce(c, op=dist.ReduceOp.MIN)
    same_stream = c[0].item() == chk and -c[1].item() == chk
    timeouts = group.timeouts() if hasattr(group, "timeouts") else 0
    if rank == 0:
        dt, ar_dt = float(t[0].item()), float(t[1].item())
        ms = dt / args.steps * 1e3
        ctx_mid = args.prompt_len + (args.warmup + args.steps / 2) * n_timed / args.steps
        rank_bytes = (streamed_weight_bytes(tw) + k * streamed_weight_bytes(dw)
                      + tcfg.kv_bytes_per_token() * ctx_mid + k * dcfg.kv_bytes_per_token() * ctx_mid)
        print(json.dumps({
            "metric": f"output tokens/sec (+ accept-len), {target} TP={world} + {draft} draft, k={k}, greedy",
            "value": round(n_timed / dt, 2), "unit": "tokens/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic (seeded random shards; random-token prompt; acceptance imposed by a fixed "
                    f"cyclic schedule at probability {args.accept_prob})",
            "config": {"workload": f"{target} target sharded over {world} ranks (heads, ffn columns, "
                                   f"vocabulary-parallel head) + {draft} draft {args.tp_draft}, k={k}, "
                                   f"batch 1, prompt {args.prompt_len} (BASELINE.json configs[3])",
                       "parallelism": f"tp{world}", "tensor_parallel_collective": args.tp_collective,
                       "draft": args.tp_draft, "hipgraph": not args.no_graph,
                       "ranks_share_one_gpu": "SD_BENCH_DEVICE" in os.environ},
            "accept_len": round(n_timed / args.steps - 1.0, 3),
            "spec_stream_equals_ar_stream": bool(c[2].item() == 1.0),
            "ranks_emit_the_same_stream": bool(same_stream), "exchange_timeouts": int(timeouts),
            "ar_tokens_per_s": round(ar_n / ar_dt, 2),
            "speedup_vs_ar": round(n_timed / dt / (ar_n / ar_dt), 3), "prefill_s": round(prefill, 3),
            "step_roofline": {"bound": "hbm", "bytes_per_step_per_rank": int(rank_bytes),
                              "achieved": round(rank_bytes / (ms * 1e-3) * 1e-9, 1), "peak": HBM_PEAK_GBS,
                              "unit": "GB/s per GPU", "frac": round(rank_bytes / (ms * 1e-3) * 1e-9 / HBM_PEAK_GBS, 4)},
            "roofline": None, "cpu_baseline": None}), flush=True)
    dist.barrier()
    del eng
    if hasattr(group, "close"):
        torch.cuda.synchronize()
        dist.barrier()
        group.close()
    dist.destroy_process_group()


def main():
    args = parse()
    if args.cpu_baseline_worker:
        cpu_baseline_worker(args.cpu_budget)
        return
    if args.tp:
        main_tp(args)
        return
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU path)")
    # Rehearsal of the multi-rank flow on a box with one GPU: SD_BENCH_BACKEND=gloo puts the
    # barrier and the reductions on the CPU, SD_BENCH_DEVICE=0 lets every rank share card 0.
    backend = os.environ.get("SD_BENCH_BACKEND", "nccl")
    local = int(os.environ.get("SD_BENCH_DEVICE", local))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    red_device = device if backend == "nccl" else torch.device("cpu")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    def barrier():
        if dist is not None:
            dist.barrier()

    from vllm_omni_draft_amd.engine import EngineConfig, SpecEngine

    k = args.k
    fp8 = args.weights == "fp8"
    ra_t, ra_d = (int(v) for v in args.read_ahead_mb.split(","))
    tw, data_kind = load_weights(args.target, 1, device, args.layers, fp8)
    dw, _ = load_weights(args.draft, 2, device, args.layers, fp8)
    g = torch.Generator().manual_seed(1234 + rank)
    prompt = torch.randint(0, min(tw.cfg.vocab, dw.cfg.vocab), (args.prompt_len,), generator=g).tolist()
    total_steps = args.warmup + args.steps
    n_truth = total_steps * (k + 1) + 8
    max_seq = args.prompt_len + n_truth + 16

    # 1) autoregressive baseline (k = 0) on the same kernels: the truth stream
    ar = SpecEngine(tw, None, EngineConfig(k=0, max_seq_len=max_seq, split_tokens=args.split_tokens,
                                           target_read_ahead_mb=ra_t,
                                           use_graph=not args.no_graph, gen_capacity=n_truth + 64))
    ovl = args.read_back == "overlapped"
    ar_tokens, ar_n, ar_dt, ar_prefill = run_generation(ar, prompt, n_truth - 4 - args.warmup,
                                                        args.warmup, barrier, ovl)
    ar_tps = ar_n / ar_dt
    truth = torch.zeros(1, max_seq, dtype=torch.int32)
    truth[0, :len(prompt)] = torch.tensor(prompt)
    truth[0, len(prompt):len(prompt) + len(ar_tokens)] = torch.tensor(ar_tokens)
    truth_dev = truth.to(device)
    del ar
    torch.cuda.empty_cache()

    # 2) speculative run
    def spec_run(alpha, steps, warmup, overlapped=ovl, draft=None):
        eng = SpecEngine(tw, draft or dw, EngineConfig(k=k, max_seq_len=max_seq, split_tokens=args.split_tokens,
                                              draft_split_tokens=args.draft_split_tokens,
                                              draft_fused_layers=args.draft_layers == "three",
                                              target_read_ahead_mb=ra_t, draft_read_ahead_mb=ra_d,
                                              use_graph=not args.no_graph, gen_capacity=n_truth + 64))
        # A fixed cyclic schedule decides how many proposals each step accepts (the timed
        # region covers whole cycles, so its accept-len is the schedule's mean, run after run);
        # the warm-up steps shift the cycle's start.
        eng.set_synthetic_acceptance(truth_dev, alpha, seed=-warmup, pattern_len=min(steps, 64))
        toks, n_timed, dt, prefill = run_generation(eng, prompt, steps, warmup, barrier, overlapped)
        n_cmp = min(len(toks), len(ar_tokens))
        exact = toks[:n_cmp] == ar_tokens[:n_cmp]
        m = eng.metrics()
        res = {"tokens": n_timed, "seconds": dt, "prefill_s": prefill, "exact": exact,
               "pattern": list(eng.accept_pattern),
               "accept_len": n_timed / steps - 1.0,          # timed steps only
               "emitted_per_step": n_timed / steps,
               "draft_acceptance_rate": m["accepted"] / max(m["draft"], 1),
               "system_efficiency": m["emitted"] / max((m["draft"] // k) * (k + 1), 1)}
        return eng, res

    eng, main_res = spec_run(args.accept_prob, args.steps, args.warmup)
    tokens, dt = main_res["tokens"], main_res["seconds"]
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=red_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        n = torch.tensor([tokens, int(main_res["exact"])], dtype=torch.float64, device=red_device)
        dist.all_reduce(n, op=dist.ReduceOp.SUM)
        dt, tokens = float(t.item()), float(n[0].item())
        all_exact = int(n[1].item()) == world
    else:
        all_exact = main_res["exact"]

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    value = tokens / dt
    ms_per_step = dt / args.steps * 1e3
    ctx_mid = args.prompt_len + (args.warmup + args.steps / 2) * main_res["emitted_per_step"]
    sb = step_bytes(tw.cfg, dw.cfg, k, ctx_mid,
                    *((streamed_weight_bytes(tw), streamed_weight_bytes(dw)) if fp8 else ()))
    roof = kernel_roofline(eng, tw, k + 1)
    attn = attention_roofline(device, k)
    draft_pass = draft_pass_roofline(dw, device, int(ctx_mid), args.draft_split_tokens or args.split_tokens,
                                     fused_layers=args.draft_layers == "three", read_ahead_mb=ra_d)
    sweep = []
    other_mode = None
    fp8_draft = None
    subset_head = {}
    if not args.no_sweep and world == 1:
        del eng
        torch.cuda.empty_cache()
        e2, r = spec_run(args.accept_prob, args.steps, args.warmup, not ovl)
        other_mode = {"read_back": "sync" if ovl else "overlapped",
                      "tokens_per_s": round(r["tokens"] / r["seconds"], 2),
                      "ms_per_step": round(r["seconds"] / args.steps * 1e3, 4), "exact": r["exact"]}
        del e2
        torch.cuda.empty_cache()
        if not fp8:
            # The draft only proposes: its precision never shows in the emitted tokens, which
            # stay the bf16 target's own greedy stream.  An fp8 copy of the draft halves the
            # bytes of its k passes.  (The imposed acceptance ignores what quantisation would
            # cost a real draft in agreement, so this is a separate data point, not `value`.)
            dw8, _ = load_weights(args.draft, 2, device, args.layers, True)
            e2, r = spec_run(args.accept_prob, args.steps, args.warmup, ovl, dw8)
            fp8_draft = {"tokens_per_s": round(r["tokens"] / r["seconds"], 2),
                         "ms_per_step": round(r["seconds"] / args.steps * 1e3, 4),
                         "speedup_vs_ar": round(r["tokens"] / r["seconds"] / ar_tps, 3),
                         "exact": r["exact"],
                         "note": "bf16 target (emitted stream identical to its autoregressive "
                                 "stream), fp8-e4m3 weight copy of the draft"}
            del e2, dw8
            torch.cuda.empty_cache()
            # The same for a draft LM head over a subset of the vocabulary (here the first 32768
            # token ids; in deployment the frequent ones): a quarter of the head's bytes per draft
            # pass.  What the restriction costs a real draft in acceptance is not in this number.
            sub = range(32768)
            for key, q8 in (("bf16_draft", False), ("fp8_draft", True)):
                dws, _ = load_weights(args.draft, 2, device, args.layers, q8, head_subset=sub)
                e2, r = spec_run(args.accept_prob, args.steps, args.warmup, ovl, dws)
                subset_head[key] = {"tokens_per_s": round(r["tokens"] / r["seconds"], 2),
                                    "ms_per_step": round(r["seconds"] / args.steps * 1e3, 4),
                                    "speedup_vs_ar": round(r["tokens"] / r["seconds"] / ar_tps, 3),
                                    "exact": r["exact"]}
                del e2, dws
                torch.cuda.empty_cache()
            subset_head["note"] = ("bf16 target (emitted stream identical to its autoregressive stream); "
                                   "the draft's LM head covers 32768 of the 128256 tokens "
                                   "(SpecDecodeConfig.draft_vocab_subset)")
        for a in (0.0, 0.6, 0.7, 0.9, 1.0):
            e2, r = spec_run(a, max(16, args.steps // 2), 4)
            sweep.append({"accept_prob": a, "tokens_per_s": round(r["tokens"] / r["seconds"], 1),
                          "accept_len": round(r["accept_len"], 3), "speedup_vs_ar": round(
                              r["tokens"] / r["seconds"] / ar_tps, 3), "exact": r["exact"]})
            del e2
            torch.cuda.empty_cache()
    grid = {}
    if not args.no_sweep and world == 1:
        # the other points of the measurement grid, a few seconds each
        bucket = args.split_tokens
        while bucket < ctx_mid + k + 2:
            bucket *= 2
        grid["verify_attention_roofline_headline"] = attention_roofline_headline(
            device, k, int(ctx_mid), args.split_tokens, bucket)
        gs, gw = max(16, args.steps // 2), 4
        grid["prompt_sweep"] = [grid_point(tw, dw, 1, n, k, gs, gw, args.accept_prob, not args.no_graph,
                                           split_tokens=args.split_tokens) for n in (128, 4096)]
        grid["batch8"] = grid_point(tw, dw, 8, args.prompt_len, k, gs, gw, args.accept_prob, not args.no_graph,
                                    split_tokens=args.split_tokens)
        t1 = grid_point(tw, dw, 1, args.prompt_len, k, gs, gw, args.accept_prob, not args.no_graph,
                        temperature=1.0, split_tokens=args.split_tokens)
        t1["note"] = ("temperature 1.0, seeded per-sequence streams, rejection sampling; acceptance is "
                      "what seeded random weights give (near 0), so this point prices the sampling path's "
                      "step (softmax, draws, rejection kernels), not a speed-up")
        grid["sampling_t1"] = t1
    cpu = None
    if not args.no_cpu_baseline and world == 1:   # reported at N = 1 only
        try:
            cpu = cpu_baseline()
        except Exception as e:  # the baseline is a report item, never the product path
            cpu = {"value": None, "unit": "tokens/s", "cores": None,
                   "kind": "port", "sample": f"failed: {type(e).__name__}: {e}"}

    line = {
        "metric": "output tokens/sec (+ accept-len), Llama-3-8B + Llama-3.2-1B draft, k=5, greedy",
        "value": round(value, 2), "unit": "tokens/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        "dtype": "fp8 weights (e4m3, per-row scales), bf16 activations, f32 accumulation" if fp8 else "bf16",
        "data": f"synthetic ({data_kind}; random-token prompt; acceptance imposed by a fixed cyclic "
                f"schedule with the per-step accepted counts independent acceptances of probability "
                f"{args.accept_prob} would give: real Llama-3-8B/1B acceptance is unknown here, no "
                f"weights on the box)",
        "config": {"workload": f"{args.target} target + {args.draft} draft, k={k}, batch 1 per GPU, "
                               f"prompt {args.prompt_len}, TP=1 (BASELINE.json configs[1])",
                   "accept_prob": args.accept_prob, "parallelism": f"dp{world} (independent replicas)",
                   "hipgraph": not args.no_graph, "split_tokens": args.split_tokens,
                   "read_back": args.read_back + (" (every step's tokens reach the host inside the "
                                                  "timed region; step t is read while step t+1 runs)"
                                                  if ovl else " (after every step)")},
        "other_read_back_mode": other_mode,
        "bf16_target_fp8_draft": fp8_draft,
        "bf16_target_draft_head_subset": subset_head or None,
        "accept_schedule": {"accepted_per_step": main_res["pattern"],
                            "mean": round(sum(main_res["pattern"]) / len(main_res["pattern"]), 4),
                            "expected_at_accept_prob": round(sum(args.accept_prob ** i for i in range(1, k + 1)), 4),
                            "note": "deterministic (sd_debug_force_proposals): ms_per_step and "
                                    "accept_len below belong together; value = (accept_len + 1) / "
                                    "ms_per_step"},
        "accept_len": round(main_res["accept_len"], 3),
        "emitted_per_step": round(main_res["emitted_per_step"], 3),
        "draft_acceptance_rate": round(main_res["draft_acceptance_rate"], 4),
        "system_efficiency": round(main_res["system_efficiency"], 4),
        "spec_stream_equals_ar_stream": bool(all_exact),
        "ar_tokens_per_s": round(ar_tps, 2), "speedup_vs_ar": round(value / world / ar_tps, 3),
        "prefill_s": round(main_res["prefill_s"], 3),
        "step_roofline": {"bound": "hbm", "bytes_per_step": int(sb),
                          "achieved": round(sb / (ms_per_step * 1e-3) * 1e-9, 1), "peak": HBM_PEAK_GBS,
                          "unit": "GB/s", "frac": round(sb / (ms_per_step * 1e-3) * 1e-9 / HBM_PEAK_GBS, 4)},
        "roofline": roof,
        "verify_attention_roofline": attn,
        "draft_pass_roofline": draft_pass,
        "verify_attention_roofline_headline": grid.get("verify_attention_roofline_headline"),
        "prompt_sweep": grid.get("prompt_sweep"),
        "batch8": grid.get("batch8"),
        "sampling_t1": grid.get("sampling_t1"),
        "accept_sweep": sweep,
        "cpu_baseline": cpu,
    }
    print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
