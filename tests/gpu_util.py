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
