"""Device-resident model weights in the layout the kernels stream: bf16,
q/k/v fused into one matrix and gate/up fused into one, every matrix re-tiled
once for the 16x16x32 MFMA (ops.tile_weights; layouts in include/sd_amd.h).
With fp8=True the layer matrices and the LM head are quantised at load to
fp8-e4m3 with one f32 scale per output row (weight-only: activations, norms,
biases and the embedding table stay bf16).

Sources: a Hugging Face style state dict (names as in transformers), a local
directory of *.safetensors shards, or seeded random initialisation.  The
reference only ever loads by hub name
(/root/reference/src/qwen_image_gen/model.py:46-65), which needs a network.
"""
from __future__ import annotations

import glob
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch

from . import ops
from .config import ModelConfig, rope_table


@dataclass
class LayerTensors:
    attn_norm_w: torch.Tensor
    qkv_w: "ops.TiledWeight"      # TILE_ROTARY
    o_w: "ops.TiledWeight"        # TILE_PLAIN
    mlp_norm_w: torch.Tensor
    up_w: "ops.TiledWeight"       # TILE_GATED (llama) / TILE_PLAIN (opt)
    down_w: "ops.TiledWeight"     # TILE_PLAIN
    attn_norm_b: Optional[torch.Tensor] = None
    qkv_b: Optional[torch.Tensor] = None
    o_b: Optional[torch.Tensor] = None
    mlp_norm_b: Optional[torch.Tensor] = None
    up_b: Optional[torch.Tensor] = None
    down_b: Optional[torch.Tensor] = None


@dataclass
class ModelWeights:
    cfg: ModelConfig
    embed: torch.Tensor           # row-major, or the tiled LM head's buffer when tied
    final_norm_w: torch.Tensor
    lm_head: "ops.TiledWeight"    # TILE_PLAIN
    layers: List[LayerTensors]
    embed_tiled: bool = False
    pos_table: Optional[torch.Tensor] = None
    final_norm_b: Optional[torch.Tensor] = None
    rope_cs: Optional[torch.Tensor] = None
    fp8: bool = False
    # LM head over a subset of the vocabulary (a draft model's cheaper head): head_map[i] is the
    # token id of head row i (int32, on the device); None = the whole vocabulary in order
    head_map: Optional[torch.Tensor] = None

    def nbytes(self) -> int:
        seen, total = set(), 0
        for t in self.tensors():
            if t.data_ptr() not in seen:
                seen.add(t.data_ptr())
                total += t.numel() * t.element_size()
        return total

    def tensors(self):
        def parts(t):
            if isinstance(t, ops.TiledWeight):
                yield t.data
                if t.scale is not None:
                    yield t.scale
            elif t is not None:
                yield t

        for t in (self.embed, self.final_norm_w, self.lm_head, self.pos_table, self.final_norm_b):
            yield from parts(t)
        for l in self.layers:
            for t in vars(l).values():
                yield from parts(t)


def _dev(t: torch.Tensor, device) -> torch.Tensor:
    return t.to(device=device, dtype=torch.bfloat16).contiguous()


def _tile_matrix(w: torch.Tensor, mode: int = ops.TILE_PLAIN, head_dim: int = 0,
                 fp8: bool = False) -> "ops.TiledWeight":
    if fp8:
        q, scale = ops.quantize_fp8(w)
        return ops.tile_weights(q, mode, head_dim, scale=scale)
    return ops.tile_weights(w, mode, head_dim)


def _head_subset(cfg: ModelConfig, head_subset, device):
    """(config whose head has one row per listed token, the ids as a device tensor)."""
    ids = torch.as_tensor(list(head_subset), dtype=torch.int64)
    if ids.numel() == 0 or ids.numel() % 16 or ids.min() < 0 or ids.max() >= cfg.vocab:
        raise ValueError("head_subset: a positive multiple of 16 token ids inside the vocabulary")
    if ids.unique().numel() != ids.numel():
        raise ValueError("head_subset: token ids must be distinct")
    if cfg.lm_vocab:
        raise ValueError("head_subset: the head is already a slice (vocabulary-parallel)")
    import dataclasses
    return dataclasses.replace(cfg, lm_vocab=int(ids.numel())), ids.to(device)


def _embed_and_head(embed: torch.Tensor, head, device, fp8: bool = False, subset=None):
    """Untied: row-major embedding + tiled head.  Tied: ONE tiled buffer serves
    both (the embedding gather reads rows out of the tiled layout); with fp8 the
    head is a quantised copy and the embedding table stays row-major bf16.  subset (token
    ids): the head keeps those rows only, in that order; the embedding table stays whole."""
    if subset is not None:
        src = embed if head is None else _dev(head, device)
        return embed, _tile_matrix(src[subset].contiguous(), fp8=fp8), False
    if head is None:
        if fp8:
            return embed, _tile_matrix(embed, fp8=True), False
        t = _tile_matrix(embed)
        return t.data, t, True
    return embed, _tile_matrix(_dev(head, device), fp8=fp8), False


def language_model_state(state: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """The language model's tensors of a Qwen2-VL style checkpoint under the names of a text
    checkpoint: "model.language_model.X" -> "model.X", the vision tower ("visual.",
    "model.visual.") dropped.  Text checkpoints pass through unchanged."""
    out = {}
    for name, t in state.items():
        if name.startswith(("visual.", "model.visual.")):
            continue
        if name.startswith("model.language_model."):
            name = "model." + name[len("model.language_model."):]
        out[name] = t
    return out


def from_state_dict(cfg: ModelConfig, state: Dict[str, torch.Tensor], device="cuda",
                    fp8: bool = False, head_subset=None) -> ModelWeights:
    """head_subset (token ids, a multiple of 16 of them): the LM head computes logits for those
    tokens only (ModelWeights.head_map; for draft models: a head over the frequent tokens costs
    a fraction of the whole one, and whatever the draft cannot propose the target still emits)."""
    cfg.validate()
    state = language_model_state(state)
    sub = None
    if head_subset is not None:
        cfg, sub = _head_subset(cfg, head_subset, device)

    def _tile(w, mode=ops.TILE_PLAIN, head_dim=0):  # every layer matrix alike
        return _tile_matrix(w, mode, head_dim, fp8=fp8)

    g = lambda n: state[n]
    opt = lambda n: state.get(n)
    layers = []
    if cfg.arch == "llama":
        pre = "model."
        for i in range(cfg.n_layers):
            lp = f"{pre}layers.{i}."
            qkv_w = torch.cat([g(lp + "self_attn.q_proj.weight"), g(lp + "self_attn.k_proj.weight"),
                               g(lp + "self_attn.v_proj.weight")], dim=0)
            qb = opt(lp + "self_attn.q_proj.bias")
            qkv_b = None if qb is None else torch.cat(
                [qb, g(lp + "self_attn.k_proj.bias"), g(lp + "self_attn.v_proj.bias")], dim=0)
            up_w = torch.cat([g(lp + "mlp.gate_proj.weight"), g(lp + "mlp.up_proj.weight")], dim=0)
            layers.append(LayerTensors(
                attn_norm_w=_dev(g(lp + "input_layernorm.weight"), device),
                qkv_w=_tile(_dev(qkv_w, device), ops.TILE_ROTARY, cfg.head_dim),
                qkv_b=None if qkv_b is None else _dev(qkv_b, device),
                o_w=_tile(_dev(g(lp + "self_attn.o_proj.weight"), device)),
                mlp_norm_w=_dev(g(lp + "post_attention_layernorm.weight"), device),
                up_w=_tile(_dev(up_w, device), ops.TILE_GATED),
                down_w=_tile(_dev(g(lp + "mlp.down_proj.weight"), device))))
        embed, lm_head, tied = _embed_and_head(_dev(g(pre + "embed_tokens.weight"), device),
                                               opt("lm_head.weight"), device, fp8, sub)
        return ModelWeights(cfg=cfg, embed=embed, embed_tiled=tied, fp8=fp8,
                            final_norm_w=_dev(g(pre + "norm.weight"), device), lm_head=lm_head,
                            layers=layers, rope_cs=rope_table(cfg, device),
                            head_map=None if sub is None else sub.to(torch.int32))
    pre = "model.decoder."
    for i in range(cfg.n_layers):
        lp = f"{pre}layers.{i}."
        qkv_w = torch.cat([g(lp + f"self_attn.{n}_proj.weight") for n in "qkv"], dim=0)
        qkv_b = torch.cat([g(lp + f"self_attn.{n}_proj.bias") for n in "qkv"], dim=0)
        layers.append(LayerTensors(
            attn_norm_w=_dev(g(lp + "self_attn_layer_norm.weight"), device),
            attn_norm_b=_dev(g(lp + "self_attn_layer_norm.bias"), device),
            qkv_w=_tile(_dev(qkv_w, device), ops.TILE_ROTARY, cfg.head_dim), qkv_b=_dev(qkv_b, device),
            o_w=_tile(_dev(g(lp + "self_attn.out_proj.weight"), device)),
            o_b=_dev(g(lp + "self_attn.out_proj.bias"), device),
            mlp_norm_w=_dev(g(lp + "final_layer_norm.weight"), device),
            mlp_norm_b=_dev(g(lp + "final_layer_norm.bias"), device),
            up_w=_tile(_dev(g(lp + "fc1.weight"), device)), up_b=_dev(g(lp + "fc1.bias"), device),
            down_w=_tile(_dev(g(lp + "fc2.weight"), device)), down_b=_dev(g(lp + "fc2.bias"), device)))
    embed, lm_head, tied = _embed_and_head(_dev(g(pre + "embed_tokens.weight"), device),
                                           opt("lm_head.weight"), device, fp8, sub)
    return ModelWeights(cfg=cfg, embed=embed, embed_tiled=tied, fp8=fp8,
                        head_map=None if sub is None else sub.to(torch.int32),
                        pos_table=_dev(g(pre + "embed_positions.weight"), device),
                        final_norm_w=_dev(g(pre + "final_layer_norm.weight"), device),
                        final_norm_b=_dev(g(pre + "final_layer_norm.bias"), device),
                        lm_head=lm_head, layers=layers)


def shard_state_dict(cfg: ModelConfig, state: Dict[str, torch.Tensor], tp_size: int,
                     tp_rank: int, vocab_parallel: bool = False):
    """(per-rank config, per-rank state dict) for tensor parallelism over `tp_size` GPUs.

    Column-parallel: q/k/v projections by head and gate/up (fc1) by ffn column keep the
    rows of the rank's share; row-parallel: o_proj and down_proj (fc2) keep the matching
    input columns, their partial products are summed over the ranks.  Norms, embeddings,
    LM head and the biases of the row-parallel layers stay whole (rank 0 applies them).
    vocab_parallel: the LM head keeps only the rank's slice of vocabulary rows (a tied head
    becomes an explicit "lm_head.weight" slice; the embedding table stays whole)."""
    if not 0 <= tp_rank < tp_size:
        raise ValueError(f"tp_rank {tp_rank} outside 0..{tp_size - 1}")
    local = cfg.tp_shard(tp_size, vocab_parallel)
    D = cfg.head_dim
    q0, q1 = tp_rank * local.n_heads * D, (tp_rank + 1) * local.n_heads * D
    k0, k1 = tp_rank * local.n_kv_heads * D, (tp_rank + 1) * local.n_kv_heads * D
    f0, f1 = tp_rank * local.ffn, (tp_rank + 1) * local.ffn
    rows = {"q_proj": (q0, q1), "k_proj": (k0, k1), "v_proj": (k0, k1), "gate_proj": (f0, f1),
            "up_proj": (f0, f1), "fc1": (f0, f1)}
    cols = {"o_proj": (q0, q1), "out_proj": (q0, q1), "down_proj": (f0, f1), "fc2": (f0, f1)}
    out: Dict[str, torch.Tensor] = {}
    for name, t in state.items():
        parts = name.split(".")
        mod, kind = (parts[-2], parts[-1]) if len(parts) >= 2 else ("", "")
        if "layers" in parts and mod in rows:
            a, b = rows[mod]
            t = t[a:b]
        elif "layers" in parts and mod in cols and kind == "weight":
            a, b = cols[mod]
            t = t[:, a:b]
        out[name] = t.contiguous()
    if vocab_parallel:
        v0, v1 = tp_rank * local.lm_vocab, (tp_rank + 1) * local.lm_vocab
        head = state.get("lm_head.weight")
        if head is None:
            head = next(t for n, t in state.items() if n.endswith("embed_tokens.weight"))
        out["lm_head.weight"] = head[v0:v1].contiguous()
    return local, out


def random_init(cfg: ModelConfig, seed: int = 0, device="cuda", std: float = 0.02,
                fp8: bool = False, head_subset=None, tp_rank: int = 0) -> ModelWeights:
    """Seeded random weights generated directly on the device in the fused
    layout (no host copy: an 8B model is 16 GB).  head_subset: as for from_state_dict (the
    other tensors are those of the same seed without it).  tp_rank: cfg is one rank's share
    of a tensor-parallel model (ModelConfig.tp_shard); every rank draws the same replicated
    tensors (norms, embedding) and its own sharded matrices (rank 0's, rotated along the input
    dimension), so that the ranks' partial products differ as real shards' do."""
    cfg.validate()
    sub = None
    full_rows = cfg.head_rows
    if head_subset is not None:
        cfg, sub = _head_subset(cfg, head_subset, device)
    gen = torch.Generator(device=device).manual_seed(seed)

    def _tile(w, mode=ops.TILE_PLAIN, head_dim=0):
        return _tile_matrix(w, mode, head_dim, fp8=fp8)

    def rn(*shape, s=std, sharded=True):
        w = (torch.randn(*shape, generator=gen, device=device, dtype=torch.float32) * s
             ).to(torch.bfloat16)
        if tp_rank and sharded and len(shape) == 2:
            w = w.roll(17 * tp_rank, dims=1)
        return w

    def ones(n):
        return (1.0 + torch.randn(n, generator=gen, device=device) * 0.05).to(torch.bfloat16)

    H, F = cfg.hidden, cfg.ffn
    layers = []
    for _ in range(cfg.n_layers):
        if cfg.arch == "llama":
            layers.append(LayerTensors(
                attn_norm_w=ones(H), qkv_w=_tile(rn(cfg.qkv_rows, H), ops.TILE_ROTARY, cfg.head_dim),
                qkv_b=rn(cfg.qkv_rows) if cfg.qkv_bias else None,
                o_w=_tile(rn(H, cfg.n_heads * cfg.head_dim)), mlp_norm_w=ones(H),
                up_w=_tile(rn(2 * F, H), ops.TILE_GATED), down_w=_tile(rn(H, F))))
        else:
            layers.append(LayerTensors(
                attn_norm_w=ones(H), attn_norm_b=rn(H),
                qkv_w=_tile(rn(cfg.qkv_rows, H), ops.TILE_ROTARY, cfg.head_dim),
                qkv_b=rn(cfg.qkv_rows), o_w=_tile(rn(H, cfg.n_heads * cfg.head_dim)), o_b=rn(H),
                mlp_norm_w=ones(H), mlp_norm_b=rn(H), up_w=_tile(rn(F, H)), up_b=rn(F),
                down_w=_tile(rn(H, F)), down_b=rn(H)))
    embed = rn(cfg.vocab, H, s=1.0 if cfg.arch == "llama" else 0.5, sharded=False)
    embed, lm_head, tied = _embed_and_head(embed, None if cfg.tie_embeddings else rn(full_rows, H),
                                           device, fp8, sub)
    w = ModelWeights(cfg=cfg, embed=embed, embed_tiled=tied, final_norm_w=ones(H), lm_head=lm_head,
                     layers=layers, fp8=fp8, head_map=None if sub is None else sub.to(torch.int32))
    if cfg.arch == "llama":
        w.rope_cs = rope_table(cfg, device)
    else:
        w.pos_table = rn(cfg.max_pos + cfg.pos_bias, H, s=0.5, sharded=False)
        w.final_norm_b = rn(H)
    return w


def from_safetensors_dir(cfg: ModelConfig, model_dir: str, device="cuda",
                         fp8: bool = False, head_subset=None) -> ModelWeights:
    """Loads every *.safetensors shard of a local Hugging Face checkpoint."""
    from safetensors import safe_open

    files = sorted(glob.glob(os.path.join(model_dir, "*.safetensors")))
    if not files:
        raise FileNotFoundError(f"no *.safetensors under {model_dir}")
    state: Dict[str, torch.Tensor] = {}
    for fn in files:
        with safe_open(fn, framework="pt", device="cpu") as f:
            for key in f.keys():
                state[key] = f.get_tensor(key)
    if cfg.arch == "opt" and "decoder.embed_tokens.weight" in state:
        state = {"model." + k if k.startswith("decoder.") else k: v for k, v in state.items()}
    return from_state_dict(cfg, state, device, fp8=fp8, head_subset=head_subset)
