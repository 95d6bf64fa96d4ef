"""Paged KV cache storage and page allocation.

Page = 16 tokens of one KV head; layouts are documented in include/sd_amd.h.
The cache is zero-filled at allocation (the attention kernel reads unwritten
slots of a partly filled page and masks them, so they must be finite).
"""
from __future__ import annotations

from typing import Dict, List

import torch

from .config import ModelConfig

PAGE = 16


class OutOfPages(RuntimeError):
    pass


class PageAllocator:
    """Free-list allocator of cache pages shared by the sequences of one model."""

    def __init__(self, num_pages: int):
        self.num_pages = num_pages
        self._free = list(range(num_pages - 1, -1, -1))

    @property
    def num_free(self) -> int:
        return len(self._free)

    def alloc(self) -> int:
        if not self._free:
            raise OutOfPages("KV cache is out of pages")
        return self._free.pop()

    def free(self, pages: List[int]) -> None:
        self._free.extend(pages)


class PagedKVCache:
    """K and V pages of every layer of one model, plus the block tables of the
    running batch as a device tensor."""

    def __init__(self, cfg: ModelConfig, num_pages: int, max_batch: int, max_seq_len: int,
                 device="cuda"):
        self.cfg = cfg
        self.num_pages = num_pages
        self.max_blocks_per_seq = (max_seq_len + PAGE - 1) // PAGE
        page_elems = cfg.head_dim * PAGE
        shape = (cfg.n_layers, num_pages, cfg.n_kv_heads, page_elems)
        self.k = torch.zeros(shape, dtype=torch.bfloat16, device=device)
        self.v = torch.zeros(shape, dtype=torch.bfloat16, device=device)
        self.layer_stride_bytes = num_pages * cfg.n_kv_heads * page_elems * 2
        self.allocator = PageAllocator(num_pages)
        self.block_tables = torch.zeros(max_batch, self.max_blocks_per_seq, dtype=torch.int32,
                                        device=device)
        self._host_tables: List[List[int]] = [[] for _ in range(max_batch)]

    def pages_of(self, slot: int) -> List[int]:
        return self._host_tables[slot]

    def ensure_capacity(self, slot: int, n_tokens: int, grow_to: int = 0) -> None:
        """Makes sure batch slot `slot` owns pages for positions 0..n_tokens-1.  When pages
        must be added anyway, grow_to > n_tokens takes pages for that many tokens at once
        (as far as free pages allow)."""
        need = (n_tokens + PAGE - 1) // PAGE
        if need > self.max_blocks_per_seq:
            raise OutOfPages(f"sequence of {n_tokens} tokens exceeds max_seq_len")
        tbl = self._host_tables[slot]
        if len(tbl) >= need:
            return
        first_new = len(tbl)
        while len(tbl) < need:
            tbl.append(self.allocator.alloc())
        want = min((grow_to + PAGE - 1) // PAGE, self.max_blocks_per_seq)
        while len(tbl) < want and self.allocator.num_free > 0:
            tbl.append(self.allocator.alloc())
        need = len(tbl)
        new = torch.tensor(tbl[first_new:], dtype=torch.int32)
        self.block_tables[slot, first_new:need].copy_(new, non_blocking=True)

    def release(self, slot: int) -> None:
        self.allocator.free(self._host_tables[slot])
        self._host_tables[slot] = []

    def move_slot(self, src: int, dst: int) -> None:
        """Batch slot `dst` (empty) takes over the pages of slot `src`."""
        if self._host_tables[dst]:
            raise ValueError(f"slot {dst} still owns pages")
        self._host_tables[dst] = self._host_tables[src]
        self._host_tables[src] = []
        self.block_tables[dst].copy_(self.block_tables[src])

    def swap_slots(self, a: int, b: int) -> None:
        """Batch slots a and b exchange their pages (either may be empty)."""
        if a == b:
            return
        self._host_tables[a], self._host_tables[b] = self._host_tables[b], self._host_tables[a]
        row = self.block_tables[a].clone()
        self.block_tables[a].copy_(self.block_tables[b])
        self.block_tables[b].copy_(row)

    @staticmethod
    def pages_for(cfg: ModelConfig, max_batch: int, max_seq_len: int) -> int:
        return max_batch * ((max_seq_len + PAGE - 1) // PAGE)

    def nbytes(self) -> int:
        return 2 * self.k.numel() * 2
