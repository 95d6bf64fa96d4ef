"""Model and engine configuration.

Configuration style mirrors the reference's: a dataclass with keyword
overrides and a ValueError on unknown keys
(/root/reference/src/qwen_image_gen/config.py:34-66).
"""
from __future__ import annotations

import json
import math
import os
import dataclasses
from dataclasses import dataclass, field, fields
from typing import Optional

import torch


@dataclass
class ModelConfig:
    arch: str = "llama"            # "llama" (Llama, Qwen2) or "opt"
    n_layers: int = 2
    hidden: int = 128
    n_heads: int = 2
    n_kv_heads: int = 2
    head_dim: int = 64
    ffn: int = 256
    vocab: int = 256
    norm_eps: float = 1e-5
    rope_theta: float = 10000.0
    rope_scaling: Optional[dict] = None
    max_pos: int = 2048
    tie_embeddings: bool = False
    qkv_bias: bool = False
    pos_bias: int = 2              # OPT learned-position offset
    name: str = "custom"
    # Vocabulary-parallel LM head (tensor parallelism): rows of the head one rank holds;
    # 0 = the whole vocabulary.  The embedding table is never split.
    lm_vocab: int = 0
    # Multimodal rotary embedding (Qwen2-VL): frequency counts of the temporal, height and
    # width position components, summing to head_dim / 2; None = ordinary rotary embedding.
    mrope_section: Optional[tuple] = None

    def validate(self) -> None:
        if self.arch not in ("llama", "opt"):
            raise ValueError(f"unknown arch {self.arch!r}")
        if self.head_dim not in (64, 128):
            raise ValueError(f"head_dim {self.head_dim} not supported (64 or 128)")
        if self.n_heads % self.n_kv_heads:
            raise ValueError("n_heads must be a multiple of n_kv_heads")
        if self.hidden % 32 or self.ffn % 32 or self.vocab % 16:
            raise ValueError("hidden and ffn must be multiples of 32, vocab a multiple of 16")
        if self.mrope_section is not None:
            sec = tuple(int(v) for v in self.mrope_section)
            if self.arch != "llama" or len(sec) != 3 or min(sec) < 0 or sum(sec) != self.head_dim // 2:
                raise ValueError(f"mrope_section {self.mrope_section} must be three counts summing "
                                 f"to head_dim/2 = {self.head_dim // 2} on a rotary model")
            self.mrope_section = sec

    @property
    def qkv_rows(self) -> int:
        return (self.n_heads + 2 * self.n_kv_heads) * self.head_dim

    def param_bytes(self) -> int:
        """bf16 bytes one forward pass streams (each matrix once; a tied head
        is read twice: as embedding rows it is negligible, as LM head in full)."""
        H, F, V = self.hidden, self.ffn, self.vocab
        per_layer = self.qkv_rows * H + H * self.n_heads * self.head_dim
        per_layer += (3 if self.arch == "llama" else 2) * F * H
        return 2 * (self.n_layers * per_layer + V * H)

    def kv_bytes_per_token(self) -> int:
        return 2 * self.n_layers * self.n_kv_heads * self.head_dim * 2

    @property
    def head_rows(self) -> int:
        """Rows of the LM head (= columns of the logits) this model instance computes."""
        return self.lm_vocab or self.vocab

    def tp_shard(self, tp_size: int, vocab_parallel: bool = False) -> "ModelConfig":
        """The per-rank view under tensor parallelism: heads and ffn columns divided,
        hidden size, vocabulary and layer count unchanged; with vocab_parallel the LM head
        is cut into tp_size row slices (each a multiple of 16 rows)."""
        if vocab_parallel and self.vocab % (16 * tp_size):
            raise ValueError(f"vocab {self.vocab} must be a multiple of 16 * tp_size for a "
                             "vocabulary-parallel head")
        if tp_size < 1 or self.n_heads % tp_size or self.n_kv_heads % tp_size:
            raise ValueError(f"tp_size {tp_size} must divide n_heads={self.n_heads} and "
                             f"n_kv_heads={self.n_kv_heads}")
        if self.ffn % (32 * tp_size):
            raise ValueError(f"ffn {self.ffn} must be a multiple of 32 * tp_size")
        return dataclasses.replace(self, n_heads=self.n_heads // tp_size,
                                   n_kv_heads=self.n_kv_heads // tp_size,
                                   ffn=self.ffn // tp_size, name=f"{self.name}/tp{tp_size}",
                                   lm_vocab=self.vocab // tp_size if vocab_parallel else 0)


def create_model_config(base: Optional[ModelConfig] = None, **overrides) -> ModelConfig:
    cfg = ModelConfig(**{f.name: getattr(base, f.name) for f in fields(ModelConfig)}) \
        if base is not None else ModelConfig()
    valid = {f.name for f in fields(ModelConfig)}
    for k, v in overrides.items():
        if k not in valid:
            raise ValueError(f"Unknown config parameter: {k}")
        setattr(cfg, k, v)
    cfg.validate()
    return cfg


_LLAMA3_SCALING = {"rope_type": "llama3", "factor": 32.0, "low_freq_factor": 1.0,
                   "high_freq_factor": 4.0, "original_max_position_embeddings": 8192}

# Shapes of the public Hugging Face configs of the models the benchmark names.
PRESETS = {
    "llama-3-8b": dict(arch="llama", n_layers=32, hidden=4096, n_heads=32, n_kv_heads=8,
                       head_dim=128, ffn=14336, vocab=128256, rope_theta=500000.0,
                       norm_eps=1e-5, max_pos=8192),
    "llama-3.2-1b": dict(arch="llama", n_layers=16, hidden=2048, n_heads=32, n_kv_heads=8,
                         head_dim=64, ffn=8192, vocab=128256, rope_theta=500000.0,
                         rope_scaling=_LLAMA3_SCALING, norm_eps=1e-5, tie_embeddings=True,
                         max_pos=8192),
    "llama-3-70b": dict(arch="llama", n_layers=80, hidden=8192, n_heads=64, n_kv_heads=8,
                        head_dim=128, ffn=28672, vocab=128256, rope_theta=500000.0,
                        norm_eps=1e-5, max_pos=8192),
    "qwen2.5-7b": dict(arch="llama", n_layers=28, hidden=3584, n_heads=28, n_kv_heads=4,
                       head_dim=128, ffn=18944, vocab=152064, rope_theta=1000000.0,
                       norm_eps=1e-6, qkv_bias=True, max_pos=8192),
    "qwen2.5-0.5b": dict(arch="llama", n_layers=24, hidden=896, n_heads=14, n_kv_heads=2,
                         head_dim=64, ffn=4864, vocab=151936, rope_theta=1000000.0,
                         norm_eps=1e-6, qkv_bias=True, tie_embeddings=True, max_pos=8192),
    # the language model of Qwen2-VL-7B (vision tower: vision.VisionConfig)
    "qwen2-vl-7b": dict(arch="llama", n_layers=28, hidden=3584, n_heads=28, n_kv_heads=4,
                        head_dim=128, ffn=18944, vocab=152064, rope_theta=1000000.0,
                        norm_eps=1e-6, qkv_bias=True, max_pos=8192, mrope_section=(16, 24, 24)),
    "opt-125m": dict(arch="opt", n_layers=12, hidden=768, n_heads=12, n_kv_heads=12,
                     head_dim=64, ffn=3072, vocab=50272, norm_eps=1e-5, tie_embeddings=True,
                     max_pos=2048),
}


def preset(name: str, **overrides) -> ModelConfig:
    if name not in PRESETS:
        raise ValueError(f"unknown preset {name!r}; known: {sorted(PRESETS)}")
    return create_model_config(None, name=name, **{**PRESETS[name], **overrides})


def from_hf_config(cfg: dict, **overrides) -> ModelConfig:
    """Builds a ModelConfig from a Hugging Face config.json dict."""
    mt = cfg.get("model_type", "llama")
    if mt == "opt":
        kw = dict(arch="opt", n_layers=cfg["num_hidden_layers"], hidden=cfg["hidden_size"],
                  n_heads=cfg["num_attention_heads"], n_kv_heads=cfg["num_attention_heads"],
                  head_dim=cfg["hidden_size"] // cfg["num_attention_heads"], ffn=cfg["ffn_dim"],
                  vocab=cfg["vocab_size"], max_pos=cfg["max_position_embeddings"],
                  tie_embeddings=cfg.get("tie_word_embeddings", True), norm_eps=1e-5)
        if cfg.get("word_embed_proj_dim", cfg["hidden_size"]) != cfg["hidden_size"]:
            raise ValueError("OPT variants with word_embed_proj_dim != hidden_size are not supported")
        if not cfg.get("do_layer_norm_before", True):
            raise ValueError("post-LN OPT variants are not supported")
    elif mt in ("llama", "qwen2", "mistral", "qwen2_vl", "qwen2_5_vl", "qwen2_vl_text",
                "qwen2_5_vl_text"):
        vl = mt.startswith("qwen2_")
        if vl and "text_config" in cfg:   # newer checkpoints nest the language model's fields
            cfg = {**cfg, **cfg["text_config"]}
        nh = cfg["num_attention_heads"]
        kw = dict(arch="llama", n_layers=cfg["num_hidden_layers"], hidden=cfg["hidden_size"],
                  n_heads=nh, n_kv_heads=cfg.get("num_key_value_heads", nh),
                  head_dim=cfg.get("head_dim") or cfg["hidden_size"] // nh,
                  ffn=cfg["intermediate_size"], vocab=cfg["vocab_size"],
                  norm_eps=cfg.get("rms_norm_eps", 1e-5), rope_theta=cfg.get("rope_theta", 10000.0),
                  rope_scaling=cfg.get("rope_scaling"),
                  max_pos=min(cfg.get("max_position_embeddings", 8192), 32768),
                  tie_embeddings=cfg.get("tie_word_embeddings", False),
                  qkv_bias=(mt == "qwen2") or vl or bool(cfg.get("attention_bias", False)))
        sc = cfg.get("rope_scaling") or {}
        if sc.get("mrope_section"):
            kw["mrope_section"] = tuple(sc["mrope_section"])
    else:
        raise ValueError(f"unsupported model_type {mt!r}")
    kw.update(overrides)
    return create_model_config(None, name=cfg.get("_name_or_path", mt), **kw)


def load_hf_config(model_dir: str, **overrides) -> ModelConfig:
    with open(os.path.join(model_dir, "config.json")) as f:
        return from_hf_config(json.load(f), **overrides)


def rope_table(cfg: ModelConfig, device="cpu") -> torch.Tensor:
    """[max_pos, head_dim] float32: cos of the D/2 rotary angles, then sin.
    Handles the "llama3" frequency rescaling of Llama-3.1/3.2 configs."""
    d = cfg.head_dim
    inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, d, 2, dtype=torch.float64) / d))
    sc = cfg.rope_scaling
    if sc:
        kind = sc.get("rope_type", sc.get("type"))
        if kind == "llama3":
            old = sc["original_max_position_embeddings"]
            lo_w, hi_w = old / sc["low_freq_factor"], old / sc["high_freq_factor"]
            wl = 2 * math.pi / inv
            smooth = (old / wl - sc["low_freq_factor"]) / (sc["high_freq_factor"] - sc["low_freq_factor"])
            scaled = torch.where(wl > lo_w, inv / sc["factor"], inv)
            blend = (1 - smooth) * inv / sc["factor"] + smooth * inv
            inv = torch.where((wl <= lo_w) & (wl >= hi_w), blend, scaled)
        elif kind not in (None, "default", "mrope"):  # mrope: plain frequencies, 3 positions
            raise ValueError(f"rope scaling {kind!r} not supported")
    ang = torch.arange(cfg.max_pos, dtype=torch.float32)[:, None] * inv.float()[None, :]
    return torch.cat([ang.cos(), ang.sin()], dim=1).contiguous().to(device)
