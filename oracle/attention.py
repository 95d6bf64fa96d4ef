"""Causal attention of new query tokens over a cached context, and the page
layouts of the KV cache, restated on the CPU.  TEST INFRASTRUCTURE ONLY.

The reference repository has no attention kernel; the operation is the
scaled-dot-product attention of the decoder models the hot path runs
(transformers/models/llama/modeling_llama.py `eager_attention_forward`), with
grouped-query head sharing and the mask "query at position p sees keys 0..p".
"""
from __future__ import annotations

import numpy as np
import torch

PAGE = 16


def verify_attention(q: torch.Tensor, k_ctx: torch.Tensor, v_ctx: torch.Tensor,
                     q_positions, scale: float) -> torch.Tensor:
    """q [T, Hq, D]; k_ctx, v_ctx [n_ctx, Hkv, D] (every cached token, the new
    ones included); q_positions[t] = absolute position of query t.
    Returns [T, Hq, D] float32."""
    q = q.float()
    k = k_ctx.float()
    v = v_ctx.float()
    T, Hq, D = q.shape
    Hkv = k.shape[1]
    G = Hq // Hkv
    out = torch.zeros(T, Hq, D, dtype=torch.float32)
    for t in range(T):
        n = int(q_positions[t]) + 1
        for hq in range(Hq):
            kh = hq // G
            s = (k[:n, kh] @ q[t, hq]) * scale
            p = torch.softmax(s, dim=0)
            out[t, hq] = p @ v[:n, kh]
    return out


def v_slot(t: int) -> int:
    """Slot of token t inside a V page row: bits 2 and 3 of t exchanged."""
    return (t & 3) | (((t >> 3) & 1) << 2) | (((t >> 2) & 1) << 3)


def unpack_pages(k_cache: np.ndarray, v_cache: np.ndarray, block_table, n_tokens: int,
                 Hkv: int, D: int):
    """Reads tokens 0..n_tokens-1 of one sequence back out of the page layouts
    (K page [D/8][16][8], V page [D][16 slots]; cache [pages][Hkv][D*16]) into
    logical [n_tokens, Hkv, D] arrays.  Written element by element on purpose:
    it is the independent statement of the layout the kernels must honour."""
    k = np.zeros((n_tokens, Hkv, D), dtype=k_cache.dtype)
    v = np.zeros((n_tokens, Hkv, D), dtype=v_cache.dtype)
    kc = k_cache.reshape(-1, Hkv, D * PAGE)
    vc = v_cache.reshape(-1, Hkv, D * PAGE)
    for pos in range(n_tokens):
        page = int(block_table[pos // PAGE])
        slot = pos % PAGE
        for hh in range(Hkv):
            for d in range(D):
                k[pos, hh, d] = kc[page, hh, ((d // 8) * PAGE + slot) * 8 + d % 8]
                v[pos, hh, d] = vc[page, hh, d * PAGE + v_slot(slot)]
    return k, v
