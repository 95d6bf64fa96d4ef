"""Time of one decode pass of the draft model (and of the target) inside a replayed hipGraph,
with and without the LM head, at batch 1:

    python experiments/draft_pass.py [model] [ctx] [q_len]

Prints microseconds per pass, per layer, and the bandwidth the layers' weights stream at.
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_omni_draft_amd import _lib, config, weights  # noqa: E402
from vllm_omni_draft_amd.kv_cache import PagedKVCache  # noqa: E402
from vllm_omni_draft_amd.model_runner import DecoderModel  # noqa: E402

dev = torch.device("cuda", 0)
lib = _lib.load()
stream = torch.cuda.Stream()


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
    return ts[len(ts) // 2]


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "llama-3.2-1b"
    ctx = int(sys.argv[2]) if len(sys.argv) > 2 else 1100
    q_len = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    max_ctx = int(os.environ.get("SD_MAX_CTX", "1280"))
    cfg = config.preset(name)
    w = weights.random_init(cfg, seed=2, device=dev)
    model = DecoderModel(w, max_ctx, int(os.environ.get("SD_SPLIT", "256")),
                         wide=os.environ.get("SD_WIDE", "0") == "1",
                         fused_pairs=os.environ.get("SD_FUSED", "0") == "1")
    if model.chain_state is not None:
        print("two linear layers per launch (sd_chain_*)", flush=True)
    if os.environ.get("SD_WIDE", "0") == "1":
        print("wide kernel family", flush=True)
    B = int(os.environ.get("SD_B", "1"))          # sequences per pass, q_len rows each
    kv = PagedKVCache(cfg, PagedKVCache.pages_for(cfg, B, max_ctx), B, max_ctx, dev)
    for b in range(B):
        kv.ensure_capacity(b, max_ctx)
    kv.k.normal_(0, 0.5)
    kv.v.normal_(0, 0.5)
    seq_lens = torch.tensor([ctx] * B, dtype=torch.int32, device=dev)
    toks = torch.arange(5, 5 + 64, dtype=torch.int32, device=dev)
    logits = torch.empty(B * q_len, cfg.vocab, dtype=torch.float32, device=dev)
    layer_bytes = sum(t.nbytes() for t in (w.layers[0].qkv_w, w.layers[0].o_w, w.layers[0].up_w,
                                           w.layers[0].down_w))
    reps = 4   # passes per graph: the average hides the graph's own launch cost
    full = graph_time(lambda: [model.forward(kv, toks, 0, 0, seq_lens, 0, B, q_len, logits, max_ctx)
                               for _ in range(reps)]) / reps
    body = graph_time(lambda: [model.forward(kv, toks, 0, 0, seq_lens, 0, B, q_len, None, max_ctx)
                               for _ in range(reps)]) / reps
    per_layer = body / cfg.n_layers
    print(f"{name} B={B} ctx={ctx} q_len={q_len} max_ctx={max_ctx}: pass {full:.1f} us, without head "
          f"{body:.1f} us, head {full - body:.1f} us ({w.lm_head.nbytes() / 1e6:.0f} MB, "
          f"{w.lm_head.nbytes() / 1e6 / max(full - body, 1e-3):.2f} TB/s), layer {per_layer:.2f} us "
          f"({layer_bytes / 1e6:.1f} MB, {layer_bytes / 1e6 / per_layer:.2f} TB/s)", flush=True)


if __name__ == "__main__":
    main()
