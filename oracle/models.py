"""Decoder language models (Llama / Qwen2 style and OPT style) with an
incremental KV cache, restated in plain torch on the CPU.  TEST
INFRASTRUCTURE ONLY.

The arithmetic follows the models the hot path serves, as implemented in the
`transformers` copy in this image (cross-checked in tests/test_oracle_models.py):
  transformers/models/llama/modeling_llama.py   (RMSNorm, rotary, GQA, SwiGLU)
  transformers/models/qwen2/modeling_qwen2.py   (same + q/k/v bias)
  transformers/models/opt/modeling_opt.py       (LayerNorm, learned positions
                                                 offset by 2, ReLU MLP, biases)
The reference repository only names such models as hub ids
(/root/reference/src/performance_enhancement/test_pipeline.py:20-26).

Weights use the Hugging Face parameter names.  `bf16_acts=True` rounds every
tensor a bf16 implementation would store (layer inputs/outputs, q/k/v, MLP
activations) to bf16 while keeping float32 accumulation, which is the
arithmetic contract of the HIP path.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch


@dataclass
class OracleConfig:
    arch: str  # "llama" (also Qwen2) or "opt"
    n_layers: int
    hidden: int
    n_heads: int
    n_kv_heads: int
    head_dim: int
    ffn: int
    vocab: int
    norm_eps: float = 1e-5
    rope_theta: float = 10000.0
    rope_scaling: Optional[dict] = None
    max_pos: int = 2048
    tie_embeddings: bool = False
    pos_bias: int = 2  # OPT: learned position rows start at 2
    # Qwen2-VL multimodal rotary embedding: frequency counts of the temporal, height and width
    # position components (transformers/models/qwen2_vl/modeling_qwen2_vl.py,
    # apply_multimodal_rotary_pos_emb); None = ordinary rotary embedding
    mrope_section: Optional[tuple] = None


def rb(x: torch.Tensor, on: bool) -> torch.Tensor:
    return x.to(torch.bfloat16).float() if on else x


def rope_inv_freq(cfg: OracleConfig) -> torch.Tensor:
    d = cfg.head_dim
    inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, d, 2, dtype=torch.float64) / d))
    sc = cfg.rope_scaling
    if sc and sc.get("rope_type", sc.get("type")) == "llama3":
        factor = sc["factor"]
        lo, hi = sc["low_freq_factor"], sc["high_freq_factor"]
        old = sc["original_max_position_embeddings"]
        wavelen = 2 * math.pi / inv
        smooth = (old / wavelen - lo) / (hi - lo)
        mid = (1 - smooth) * inv / factor + smooth * inv
        out = torch.where(wavelen > old / lo, inv / factor, inv)
        is_mid = (wavelen <= old / lo) & (wavelen >= old / hi)
        inv = torch.where(is_mid, mid, out)
    return inv.float()


def rope_table(cfg: OracleConfig) -> torch.Tensor:
    """[max_pos, D] float32: cos for the D/2 frequencies, then sin."""
    inv = rope_inv_freq(cfg)
    ang = torch.arange(cfg.max_pos, dtype=torch.float32)[:, None] * inv[None, :]
    return torch.cat([ang.cos(), ang.sin()], dim=1)


class KVCache:
    """Logical cache: per layer, K and V of every token fed so far."""

    def __init__(self, n_layers: int):
        self.k: List[Optional[torch.Tensor]] = [None] * n_layers
        self.v: List[Optional[torch.Tensor]] = [None] * n_layers

    def __len__(self) -> int:
        return 0 if self.k[0] is None else self.k[0].shape[0]

    def truncate(self, n: int) -> None:
        for i in range(len(self.k)):
            if self.k[i] is not None:
                self.k[i] = self.k[i][:n]
                self.v[i] = self.v[i][:n]

    def write(self, layer: int, pos0: int, k: torch.Tensor, v: torch.Tensor) -> None:
        if self.k[layer] is None:
            assert pos0 == 0
            self.k[layer], self.v[layer] = k.clone(), v.clone()
        else:
            self.k[layer] = torch.cat([self.k[layer][:pos0], k], dim=0)
            self.v[layer] = torch.cat([self.v[layer][:pos0], v], dim=0)


class OracleLM:
    def __init__(self, cfg: OracleConfig, state: Dict[str, torch.Tensor], bf16_acts: bool = False,
                 compute_dtype: torch.dtype = torch.float32):
        self.cfg = cfg
        self.bf = bf16_acts
        self.dt = compute_dtype
        self.w = {k: v.to(compute_dtype) for k, v in state.items()}
        self.rope = rope_table(cfg) if cfg.arch == "llama" else None

    def new_cache(self) -> KVCache:
        return KVCache(self.cfg.n_layers)

    # -- pieces -----------------------------------------------------------------
    def _lin(self, x, name, bias=True):
        y = torch.nn.functional.linear(x.to(self.dt), self.w[name + ".weight"]).float()
        b = self.w.get(name + ".bias") if bias else None
        if b is not None:
            y = y + b.float()
        return y

    def _rms(self, x, name):
        v = x.float()
        r = torch.rsqrt(v.pow(2).mean(-1, keepdim=True) + self.cfg.norm_eps)
        return rb(rb(v * r, self.bf) * self.w[name + ".weight"].float(), self.bf)

    def _ln(self, x, name):
        y = torch.nn.functional.layer_norm(x.float(), (x.shape[-1],), self.w[name + ".weight"].float(),
                                           self.w[name + ".bias"].float(), self.cfg.norm_eps)
        return rb(y, self.bf)

    def _attend(self, q, layer, cache, positions):
        from .attention import verify_attention
        scale = self.cfg.head_dim ** -0.5
        return verify_attention(q, cache.k[layer], cache.v[layer], positions, scale)

    # -- forward ----------------------------------------------------------------
    def forward(self, tokens, pos0: int, cache: KVCache, rope_pos=None,
                input_embeds=None) -> torch.Tensor:
        """Feeds `tokens` at positions pos0.. (their K/V replace anything cached
        from pos0 on) and returns float32 logits [len(tokens), vocab].
        rope_pos ([3, T] integers; models with mrope_section): the temporal, height and width
        rotary positions of the tokens instead of pos0..; input_embeds ([T, hidden]): rows that
        replace the embedding lookup (image embeddings merged in)."""
        cfg, bf = self.cfg, self.bf
        tok = torch.as_tensor(tokens, dtype=torch.long)
        T = tok.shape[0]
        pos = torch.arange(pos0, pos0 + T)
        Hq, Hkv, D = cfg.n_heads, cfg.n_kv_heads, cfg.head_dim
        if cfg.arch == "llama":
            pre = "model."
            if input_embeds is not None:
                h = torch.as_tensor(input_embeds).float()
            else:
                h = self.w[pre + "embed_tokens.weight"][tok].float()
        else:
            pre = "model.decoder."
            h = self.w[pre + "embed_tokens.weight"][tok].float() + \
                self.w[pre + "embed_positions.weight"][pos + cfg.pos_bias].float()
        h = rb(h, bf)
        for li in range(cfg.n_layers):
            lp = f"{pre}layers.{li}."
            if cfg.arch == "llama":
                x = self._rms(h, lp + "input_layernorm")
                q = self._lin(x, lp + "self_attn.q_proj")
                k = self._lin(x, lp + "self_attn.k_proj")
                v = self._lin(x, lp + "self_attn.v_proj")
            else:
                x = self._ln(h, lp + "self_attn_layer_norm")
                q = self._lin(x, lp + "self_attn.q_proj")
                k = self._lin(x, lp + "self_attn.k_proj")
                v = self._lin(x, lp + "self_attn.v_proj")
            q = rb(q, bf).view(T, Hq, D)
            k = rb(k, bf).view(T, Hkv, D)
            v = rb(v, bf).view(T, Hkv, D)
            if cfg.arch == "llama":
                if rope_pos is not None:
                    # frequency i takes the position component of its section
                    rp = torch.as_tensor(rope_pos, dtype=torch.long)
                    comp = torch.repeat_interleave(torch.arange(3), torch.tensor(cfg.mrope_section))
                    idx = torch.arange(D // 2)
                    per_freq = rp[comp].T.clamp(0, cfg.max_pos - 1)          # [T, D/2]
                    cs = torch.cat([self.rope[per_freq, idx], self.rope[per_freq, idx + D // 2]], dim=1)
                else:
                    cs = self.rope[pos]  # [T, D]
                c, s = cs[:, None, : D // 2], cs[:, None, D // 2:]

                def rot(u):
                    lo, hi = u[..., : D // 2], u[..., D // 2:]
                    return torch.cat([lo * c - hi * s, hi * c + lo * s], dim=-1)

                q, k = rb(rot(q), bf), rb(rot(k), bf)
            cache.write(li, pos0, k, v)
            a = rb(self._attend(q, li, cache, pos), bf).reshape(T, Hq * D)
            if cfg.arch == "llama":
                h = rb(rb(self._lin(a, lp + "self_attn.o_proj"), bf) + h, bf)
                x = self._rms(h, lp + "post_attention_layernorm")
                g = rb(self._lin(x, lp + "mlp.gate_proj"), bf)
                u = rb(self._lin(x, lp + "mlp.up_proj"), bf)
                act = rb(rb(torch.nn.functional.silu(g), bf) * u, bf)
                h = rb(rb(self._lin(act, lp + "mlp.down_proj"), bf) + h, bf)
            else:
                h = rb(rb(self._lin(a, lp + "self_attn.out_proj"), bf) + h, bf)
                x = self._ln(h, lp + "final_layer_norm")
                act = rb(torch.relu(rb(self._lin(x, lp + "fc1"), bf)), bf)
                h = rb(rb(self._lin(act, lp + "fc2"), bf) + h, bf)
        if cfg.arch == "llama":
            x = self._rms(h, pre + "norm")
        else:
            x = self._ln(h, pre + "final_layer_norm")
        head = "lm_head.weight" if "lm_head.weight" in self.w else pre + "embed_tokens.weight"
        return torch.nn.functional.linear(x.to(self.dt), self.w[head]).float()


def random_state(cfg: OracleConfig, seed: int, std: float = 0.05, qkv_bias: bool = False,
                 dtype: torch.dtype = torch.float32) -> Dict[str, torch.Tensor]:
    """Seeded random weights under the Hugging Face parameter names."""
    g = torch.Generator().manual_seed(seed)

    def rn(*shape, s=std):
        return (torch.randn(*shape, generator=g) * s).to(dtype)

    H, D, Hq, Hkv, F = cfg.hidden, cfg.head_dim, cfg.n_heads, cfg.n_kv_heads, cfg.ffn
    st: Dict[str, torch.Tensor] = {}
    if cfg.arch == "llama":
        st["model.embed_tokens.weight"] = rn(cfg.vocab, H, s=1.0)
        for i in range(cfg.n_layers):
            lp = f"model.layers.{i}."
            st[lp + "input_layernorm.weight"] = (1 + rn(H)).to(dtype)
            st[lp + "post_attention_layernorm.weight"] = (1 + rn(H)).to(dtype)
            st[lp + "self_attn.q_proj.weight"] = rn(Hq * D, H)
            st[lp + "self_attn.k_proj.weight"] = rn(Hkv * D, H)
            st[lp + "self_attn.v_proj.weight"] = rn(Hkv * D, H)
            st[lp + "self_attn.o_proj.weight"] = rn(H, Hq * D)
            if qkv_bias:
                st[lp + "self_attn.q_proj.bias"] = rn(Hq * D)
                st[lp + "self_attn.k_proj.bias"] = rn(Hkv * D)
                st[lp + "self_attn.v_proj.bias"] = rn(Hkv * D)
            st[lp + "mlp.gate_proj.weight"] = rn(F, H)
            st[lp + "mlp.up_proj.weight"] = rn(F, H)
            st[lp + "mlp.down_proj.weight"] = rn(H, F)
        st["model.norm.weight"] = (1 + rn(H)).to(dtype)
        if not cfg.tie_embeddings:
            st["lm_head.weight"] = rn(cfg.vocab, H)
    else:
        pre = "model.decoder."
        st[pre + "embed_tokens.weight"] = rn(cfg.vocab, H, s=1.0)
        st[pre + "embed_positions.weight"] = rn(cfg.max_pos + cfg.pos_bias, H, s=0.5)
        for i in range(cfg.n_layers):
            lp = f"{pre}layers.{i}."
            for nm in ("self_attn_layer_norm", "final_layer_norm"):
                st[lp + nm + ".weight"] = (1 + rn(H)).to(dtype)
                st[lp + nm + ".bias"] = rn(H)
            for nm, (o, i_) in {"self_attn.q_proj": (Hq * D, H), "self_attn.k_proj": (Hkv * D, H),
                                "self_attn.v_proj": (Hkv * D, H), "self_attn.out_proj": (H, Hq * D),
                                "fc1": (F, H), "fc2": (H, F)}.items():
                st[lp + nm + ".weight"] = rn(o, i_)
                st[lp + nm + ".bias"] = rn(o)
        st[pre + "final_layer_norm.weight"] = (1 + rn(H)).to(dtype)
        st[pre + "final_layer_norm.bias"] = rn(H)
        if not cfg.tie_embeddings:
            st["lm_head.weight"] = rn(cfg.vocab, H)
    return st
