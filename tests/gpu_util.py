"""Helpers shared by the GPU tests (not collected by pytest)."""
import numpy as np
import torch

from oracle.attention import v_slot

PAGE = 16


def pack_cache(k, v, block_table, num_pages):
    """Logical [n, Hkv, D] bf16 tensors -> page-layout caches [num_pages, Hkv, D*16]
    (K page [D/8][16][8], V page [D][16 slots]); written with index arithmetic that
    follows include/sd_amd.h, independently of the kernels."""
    n, Hkv, D = k.shape
    kc = torch.zeros(num_pages, Hkv, D * PAGE, dtype=torch.bfloat16)
    vc = torch.zeros(num_pages, Hkv, D * PAGE, dtype=torch.bfloat16)
    d = torch.arange(D)
    for pos in range(n):
        page, slot = int(block_table[pos // PAGE]), pos % PAGE
        kc[page, :, (d // 8) * 128 + slot * 8 + d % 8] = k[pos]
        vc[page, :, d * PAGE + v_slot(slot)] = v[pos]
    return kc, vc


def rb(x):
    return x.to(torch.bfloat16).float()


def pack_cache_fast(kc, vc, k, v, block_table):
    """pack_cache for long sequences: the same page layout written by one indexed assignment
    per tensor into existing caches kc, vc [num_pages, Hkv, D*16] (checked against pack_cache
    in tests/test_host_logic.py)."""
    n, Hkv, D = k.shape
    pos = torch.arange(n)
    page = block_table.long()[pos // PAGE]                       # [n]
    slot = pos % PAGE
    vslot = (slot & 3) | (((slot >> 3) & 1) << 2) | (((slot >> 2) & 1) << 3)
    d = torch.arange(D)
    k_idx = (d // 8)[None, :] * 128 + slot[:, None] * 8 + (d % 8)[None, :]     # [n, D]
    v_idx = d[None, :] * PAGE + vslot[:, None]
    h = torch.arange(Hkv)
    kc[page[:, None, None], h[None, :, None], k_idx[:, None, :]] = k
    vc[page[:, None, None], h[None, :, None], v_idx[:, None, :]] = v
