"""Host-side preparation of prompts that carry images (the Qwen2-VL language model as the
target of speculative decoding): rotary positions of the multimodal rotary embedding and the
target's input rows with the image embeddings merged in.

The arithmetic follows the `transformers` implementation in this image
(transformers/models/qwen2_vl/modeling_qwen2_vl.py: Qwen2VLModel.get_rope_index,
get_vision_position_ids, and the masked_scatter of image embeddings in Qwen2VLModel.forward);
the reference repository only names such models as hub ids
(/root/reference/src/qwen_image/v1/worker.py:28-49).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch

# placeholder token of Qwen2-VL / Qwen2.5-VL tokenizers (<|image_pad|>)
QWEN2_VL_IMAGE_TOKEN_ID = 151655


def image_token_count(grid_thw: Sequence[int], spatial_merge: int = 2) -> int:
    """Language-model tokens of one image (or video) whose vision patch grid is (t, h, w)."""
    t, h, w = (int(v) for v in grid_thw)
    if h % spatial_merge or w % spatial_merge:
        raise ValueError(f"grid {tuple(grid_thw)} is not a multiple of the merge size {spatial_merge}")
    return t * (h // spatial_merge) * (w // spatial_merge)


def mrope_positions(token_ids: Sequence[int], image_token_id: int,
                    grids_thw: Sequence[Sequence[int]], spatial_merge: int = 2
                    ) -> Tuple[torch.Tensor, int]:
    """Rotary positions [3, n] (temporal, height, width; int32) of a prompt and the delta that
    turns the cache position of every later token into its rotary position.

    Text tokens count up on all three components.  The tokens of an image whose patch grid is
    (t, h, w) take temporal position start + frame, height start + row, width start + column
    of the merged grid; the text after it continues at start + max(h, w) / merge, as the image
    path of transformers' Qwen2VLModel.get_rope_index in this image does (checked bit for bit in
    tests/test_multimodal.py).  That rule ignores the temporal extent, so a grid with more frames
    than merged rows and columns (a video clip, whose positions older transformers releases
    continued one past the largest of the three components) is refused rather than guessed."""
    ids = list(token_ids)
    n = len(ids)
    pos = torch.zeros(3, n, dtype=torch.int32)
    grids = [tuple(int(v) for v in g) for g in grids_thw]
    cur, i, gi = 0, 0, 0
    while i < n:
        if ids[i] != image_token_id:
            pos[:, i] = cur
            cur += 1
            i += 1
            continue
        if gi >= len(grids):
            raise ValueError("more runs of image tokens than image grids")
        t, h, w = grids[gi]
        gi += 1
        cnt = image_token_count((t, h, w), spatial_merge)
        if ids[i:i + cnt] != [image_token_id] * cnt:
            raise ValueError(f"image {gi - 1} needs {cnt} consecutive placeholder tokens at {i}")
        gh, gw = h // spatial_merge, w // spatial_merge
        tt, hh, ww = torch.meshgrid(torch.arange(t), torch.arange(gh), torch.arange(gw), indexing="ij")
        pos[0, i:i + cnt] = (tt.reshape(-1) + cur).int()
        pos[1, i:i + cnt] = (hh.reshape(-1) + cur).int()
        pos[2, i:i + cnt] = (ww.reshape(-1) + cur).int()
        if t > max(gh, gw):
            raise ValueError(f"grid {(t, h, w)}: {t} frames exceed the merged grid's extent "
                             f"{max(gh, gw)}; video clips are not supported")
        cur += max(gh, gw)
        i += cnt
    if gi != len(grids):
        raise ValueError("fewer runs of image tokens than image grids")
    delta = (int(pos.max().item()) + 1 - n) if n else 0
    return pos, delta


def merge_image_embeds(token_embeds: torch.Tensor, token_ids: Sequence[int], image_token_id: int,
                       image_embeds: torch.Tensor) -> torch.Tensor:
    """token_embeds [n, H] with the rows of the placeholder tokens replaced, in order, by
    image_embeds [number of placeholders, H]."""
    mask = torch.tensor([t == image_token_id for t in token_ids], dtype=torch.bool,
                        device=token_embeds.device)
    if int(mask.sum().item()) != image_embeds.shape[0]:
        raise ValueError(f"{int(mask.sum().item())} placeholder tokens but {image_embeds.shape[0]} "
                         "image embedding rows")
    out = token_embeds.clone()
    out[mask] = image_embeds.to(device=out.device, dtype=out.dtype)
    return out


def prepare_prompt(token_ids: Sequence[int], target_weights, image_embeds: Optional[torch.Tensor],
                   grids_thw: Sequence[Sequence[int]],
                   image_token_id: int = QWEN2_VL_IMAGE_TOKEN_ID, spatial_merge: int = 2
                   ) -> Tuple[torch.Tensor, torch.Tensor]:
    """(input_embeds [n, H], rope_positions [3, n]) for SpecEngine.add_request.
    target_weights: the target's ModelWeights (its token embedding table is looked up on the
    device); image_embeds: the vision tower's output rows for all images of the prompt,
    concatenated ([placeholders, H])."""
    from . import ops
    cfg = target_weights.cfg
    dev = target_weights.embed.device
    ids = torch.tensor(list(token_ids), dtype=torch.int32, device=dev)
    rows = ops.embed_tokens(ids, target_weights.embed, bool(target_weights.embed_tiled), cfg.hidden,
                            cfg.vocab)
    if image_embeds is not None:
        rows = merge_image_embeds(rows, token_ids, image_token_id, image_embeds)
    pos, _ = mrope_positions(token_ids, image_token_id, grids_thw, spatial_merge)
    return rows, pos
