"""One eager decode pass of the draft (Llama-3.2-1B shapes, 1 row) and one verify pass of the
target (Llama-3-8B shapes, 6 rows) over 1089 cached tokens, for per-kernel counter collection:
every layer has its own weights, so each launch streams from memory as in the step.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -o pmc -- python3 experiments/pmc_passes.py
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d out -o pmc -- python3 experiments/pmc_passes.py

The passes are bracketed by sd_fill_uniforms launches (a kernel the passes never use), so that a
reader of the dispatch list finds them: [marker] draft pass [marker] target pass [marker].
SD_PMC_LAYERS limits the layer count of both models (default 8).
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vllm_omni_draft_amd import config, ops, weights  # noqa: E402
from vllm_omni_draft_amd.kv_cache import PagedKVCache  # noqa: E402
from vllm_omni_draft_amd.model_runner import DecoderModel  # noqa: E402

dev = "cuda"
layers = int(os.environ.get("SD_PMC_LAYERS", "8"))
ctx = 1089
marker_out = torch.empty(64, dtype=torch.float32, device=dev)
marker_ctr = torch.zeros(1, dtype=torch.int64, device=dev)


def marker():
    ops.fill_uniforms(marker_out, 1, marker_ctr)


def one_pass(name, seed, rows):
    cfg = config.preset(name, n_layers=layers)
    w = weights.random_init(cfg, seed=seed, device=dev)
    max_ctx = ctx + 64
    model = DecoderModel(w, max_ctx, 256)
    kv = PagedKVCache(cfg, PagedKVCache.pages_for(cfg, 1, max_ctx), 1, max_ctx, dev)
    kv.ensure_capacity(0, max_ctx)
    kv.k.normal_(0, 0.5)
    kv.v.normal_(0, 0.5)
    seq_lens = torch.tensor([ctx], dtype=torch.int32, device=dev)
    toks = torch.arange(5, 5 + rows, dtype=torch.int32, device=dev)
    logits = torch.empty(rows, cfg.head_rows, dtype=torch.float32, device=dev)
    flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    flush.fill_(1)
    torch.cuda.synchronize()
    marker()
    model.forward(kv, toks, 0, 0, seq_lens, 0, 1, rows, logits, max_ctx)
    torch.cuda.synchronize()
    c = cfg
    per_layer = {
        "qkv": (c.n_heads + 2 * c.n_kv_heads) * c.head_dim * c.hidden * 2,
        "attention": 2 * c.n_kv_heads * c.head_dim * (ctx + rows) * 2,
        "o_proj": c.hidden * c.n_heads * c.head_dim * 2,
        "gate_up": 2 * c.ffn * c.hidden * 2,
        "down": c.hidden * c.ffn * 2,
    }
    print(name, "rows", rows, "layers", layers, "algorithmic bytes per launch:", per_layer,
          "lm_head", w.lm_head.nbytes())


one_pass("llama-3.2-1b", 2, 1)
one_pass("llama-3-8b", 1, 6)
marker()
torch.cuda.synchronize()
