"""The Qwen2-VL vision tower for prompts with images: pixel patches -> the rows that take the
placeholder tokens' places in the target's input (multimodal.prepare_prompt).

It runs once per image before the prompt is prefilled, outside the draft/verify loop, so it
is written with plain tensor operations (GEMMs through the BLAS library) rather than with
kernels of its own.  The arithmetic follows the `transformers` implementation in this image
(transformers/models/qwen2_vl/modeling_qwen2_vl.py: Qwen2VisionTransformerPretrainedModel,
PatchEmbed, VisionAttention, Qwen2VLVisionBlock, PatchMerger; patch order of
transformers/models/qwen2_vl/image_processing_qwen2_vl.py).  The reference repository names
the model as a hub id only (/root/reference/src/qwen_image/v1/worker.py:28-49).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Sequence

import torch
import torch.nn.functional as F


@dataclass
class VisionConfig:
    depth: int = 32
    embed_dim: int = 1280
    num_heads: int = 16
    mlp_ratio: float = 4.0
    hidden_size: int = 3584        # width of the language model the merger projects to
    in_channels: int = 3
    patch_size: int = 14
    temporal_patch_size: int = 2
    spatial_merge_size: int = 2
    rope_theta: float = 10000.0

    @staticmethod
    def from_hf_config(cfg: dict) -> "VisionConfig":
        v = cfg.get("vision_config", cfg)
        keys = ("depth", "embed_dim", "num_heads", "mlp_ratio", "hidden_size", "in_channels",
                "patch_size", "temporal_patch_size", "spatial_merge_size")
        return VisionConfig(**{k: v[k] for k in keys if k in v})


def patchify(image: torch.Tensor, cfg: VisionConfig):
    """image [C, H, W] (already resized and normalised; H, W multiples of patch_size *
    spatial_merge_size) -> (patches [h*w, C*tp*p*p], grid (1, h, w)).  The still image is
    repeated over the temporal patch; patches come in blocks of merge x merge neighbours,
    the order the merger expects."""
    C, H, W = image.shape
    p, m, tp = cfg.patch_size, cfg.spatial_merge_size, cfg.temporal_patch_size
    if H % (p * m) or W % (p * m):
        raise ValueError(f"image {H}x{W} is not a multiple of {p * m}")
    h, w = H // p, W // p
    x = image[None].expand(tp, C, H, W).reshape(1, tp, C, h // m, m, p, w // m, m, p)
    x = x.permute(0, 3, 6, 4, 7, 2, 1, 5, 8)           # t, h/m, w/m, m, m, C, tp, p, p
    return x.reshape(h * w, C * tp * p * p).contiguous(), (1, h, w)


def _rotary_angles(grids: Sequence[Sequence[int]], cfg: VisionConfig, device) -> torch.Tensor:
    """[tokens, head_dim/2] angles: the row index drives the first half of the frequencies,
    the column index the second, tokens in the patch order of `patchify`."""
    m = cfg.spatial_merge_size
    dim = cfg.embed_dim // cfg.num_heads // 2
    inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, dim, 2, dtype=torch.float32, device=device) / dim))
    out = []
    for t, h, w in grids:
        hp, wp = torch.meshgrid(torch.arange(h, device=device), torch.arange(w, device=device),
                                indexing="ij")
        blk = (h // m, m, w // m, m)
        hp = hp.reshape(blk).transpose(1, 2).flatten()
        wp = wp.reshape(blk).transpose(1, 2).flatten()
        ang = torch.cat([hp[:, None].float() * inv, wp[:, None].float() * inv], dim=1)
        out.append(ang.repeat(t, 1))
    return torch.cat(out)


class VisionTower:
    """Holds the tower's tensors (names of the checkpoint, "visual." / "model.visual." prefix
    stripped) and computes image rows."""

    def __init__(self, cfg: VisionConfig, state: Dict[str, torch.Tensor], device="cuda",
                 dtype=torch.bfloat16):
        self.cfg = cfg
        self.dtype = dtype
        self.w = {}
        for name, t in state.items():
            for pre in ("model.visual.", "visual."):
                if name.startswith(pre):
                    name = name[len(pre):]
                    break
            else:
                continue
            self.w[name] = t.to(device=device, dtype=dtype)
        need = ["patch_embed.proj.weight", "merger.ln_q.weight", "merger.mlp.0.weight",
                "merger.mlp.2.weight", f"blocks.{cfg.depth - 1}.mlp.fc2.weight"]
        missing = [n for n in need if n not in self.w]
        if missing:
            raise ValueError(f"vision tower tensors missing from the checkpoint: {missing}")

    def _lin(self, x, name):
        return F.linear(x, self.w[name + ".weight"], self.w.get(name + ".bias"))

    def _ln(self, x, name):
        return F.layer_norm(x, (x.shape[-1],), self.w[name + ".weight"], self.w[name + ".bias"], 1e-6)

    @torch.no_grad()
    def forward(self, patches: torch.Tensor, grids: Sequence[Sequence[int]]) -> torch.Tensor:
        """patches [sum t*h*w, C*tp*p*p] of all images (patchify order), grids their (t, h, w)
        -> [sum t*h*w / merge^2, hidden_size] rows for the language model."""
        cfg = self.cfg
        grids = [tuple(int(v) for v in g) for g in grids]
        n = sum(t * h * w for t, h, w in grids)
        if patches.shape[0] != n:
            raise ValueError(f"{patches.shape[0]} patches for grids that hold {n}")
        dev = self.w["patch_embed.proj.weight"].device
        x = F.linear(patches.to(device=dev, dtype=self.dtype),
                     self.w["patch_embed.proj.weight"].reshape(cfg.embed_dim, -1))
        ang = _rotary_angles(grids, cfg, dev)
        cos = torch.cat([ang, ang], dim=1).cos()[:, None, :]
        sin = torch.cat([ang, ang], dim=1).sin()[:, None, :]
        # attention stays inside one frame of one image
        frames = [h * w for t, h, w in grids for _ in range(t)]
        nh, hd = cfg.num_heads, cfg.embed_dim // cfg.num_heads

        def rot(u):
            u = u.float()
            half = hd // 2
            r = torch.cat([-u[..., half:], u[..., :half]], dim=-1)
            return (u * cos + r * sin).to(self.dtype)

        for i in range(cfg.depth):
            bp = f"blocks.{i}."
            qkv = self._lin(self._ln(x, bp + "norm1"), bp + "attn.qkv").view(n, 3, nh, hd)
            q, k, v = rot(qkv[:, 0]), rot(qkv[:, 1]), qkv[:, 2]
            outs, s0 = [], 0
            for ln in frames:
                sl = slice(s0, s0 + ln)
                o = F.scaled_dot_product_attention(q[sl].transpose(0, 1)[None], k[sl].transpose(0, 1)[None],
                                                   v[sl].transpose(0, 1)[None])
                outs.append(o[0].transpose(0, 1).reshape(ln, nh * hd))
                s0 += ln
            x = x + self._lin(torch.cat(outs), bp + "attn.proj")
            y = self._lin(self._ln(x, bp + "norm2"), bp + "mlp.fc1")
            y = y * torch.sigmoid(1.702 * y)                       # quick GELU
            x = x + self._lin(y, bp + "mlp.fc2")
        merged = self._ln(x, "merger.ln_q").view(-1, cfg.embed_dim * cfg.spatial_merge_size ** 2)
        return self._lin(F.gelu(self._lin(merged, "merger.mlp.0")), "merger.mlp.2")
