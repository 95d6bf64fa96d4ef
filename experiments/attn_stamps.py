"""In-kernel phase times of one attention workgroup (needs a build with -DSD_ATTN_STAMPS):
    make clean && make EXTRA=-DSD_ATTN_STAMPS && python experiments/attn_stamps.py
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_omni_draft_amd import _lib, ops  # noqa: E402

lib = _lib.load()
dev = "cuda"


def case(B, ctx, Hq, Hkv, D, q_len, split, max_ctx):
    pages_per_seq = (max_ctx + 15) // 16
    n_pages = B * pages_per_seq
    kc = (torch.randn(n_pages, Hkv, D * 16, device=dev) * 0.5).to(torch.bfloat16)
    vc = (torch.randn(n_pages, Hkv, D * 16, device=dev) * 0.5).to(torch.bfloat16)
    bt = torch.randperm(n_pages, device=dev).to(torch.int32).view(B, pages_per_seq).contiguous()
    q = torch.randn(B * q_len, Hq * D, device=dev).to(torch.bfloat16)
    sl = torch.full((B,), ctx, dtype=torch.int32, device=dev)
    flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    rows = []
    for it in range(6):
        flush.fill_(it)            # push K/V out of the caches
        torch.cuda.synchronize()
        ops.paged_attention(q, kc, vc, bt, sl, 0, B, q_len, Hq, Hkv, D, D ** -0.5, max_ctx, split)
        torch.cuda.synchronize()
        st = (C.c_ulonglong * 16)()
        assert lib.sd_debug_read_stamps(st) == 0
        rows.append([(st[i + 1] - st[i]) * 10 for i in range(6)])  # 100 MHz clock -> ns
    rows.sort(key=sum)
    med = rows[len(rows) // 2]
    names = ["seq_len", "q frags", "issue ring", "stream+mfma", "lds publish", "merge+store"]
    print(f"D={D} q_len={q_len} ctx={ctx} split={split}: total {sum(med)} ns  " +
          "  ".join(f"{n}={v}" for n, v in zip(names, med)), flush=True)


if __name__ == "__main__":
    lib.sd_debug_read_stamps.argtypes = [C.c_void_p]
    case(1, 200, 32, 8, 64, 1, 256, 256)
    case(1, 200, 32, 8, 128, 6, 256, 256)
    case(1, 1100, 32, 8, 64, 1, 256, 1280)
    case(1, 1100, 32, 8, 128, 6, 256, 1280)
    # the KV-bound roofline case: why are 6 query tokens slower than 1?
    case(32, 4096, 32, 8, 128, 1, 4096, 4102)
    case(32, 4096, 32, 8, 128, 6, 4096, 4102)
