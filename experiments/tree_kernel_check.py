"""Kernel level: rows of a tree pass against the same rows laid out as a chain (bitwise)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_omni_draft_amd import ops
from tests.gpu_util import pack_cache_fast
from oracle import tree as otree

Hq, Hkv, D, split = 32, 8, 128, 256
par = [0, 0, 1, 2, 3, 4, 0, 6, 7, 8, 9, 0, 11, 12, 13, 14]
masks = otree.ancestor_masks(par)
q_len = 16
worst = {}
for root in list(range(1, 120)) + [250, 255, 256, 260, 1024, 1044, 1076, 1050]:
    g = torch.Generator().manual_seed(root)
    n = root + q_len
    pages = 2048 // 16
    bt = torch.arange(pages, dtype=torch.int32).view(1, -1)
    k = torch.randn(n, Hkv, D, generator=g).to(torch.bfloat16)
    v = torch.randn(n, Hkv, D, generator=g).to(torch.bfloat16)
    q = torch.randn(q_len, Hq, D, generator=g).to(torch.bfloat16)
    kc = torch.zeros(pages, Hkv, D * 16, dtype=torch.bfloat16); vc = torch.zeros_like(kc)
    pack_cache_fast(kc, vc, k, v, bt[0])
    sl = torch.tensor([root], dtype=torch.int32).cuda()
    tm = torch.tensor(masks, dtype=torch.int64).cuda()
    out = ops.paged_attention_tree(q.reshape(q_len, Hq * D).cuda(), kc.cuda(), vc.cuda(), bt.cuda(), sl, 0, tm, 1,
                                   q_len, Hq, Hkv, D, D ** -0.5, 2048, split).cpu().view(q_len, Hq, D)
    for chain in ([1, 2, 3, 4, 5], [6, 7, 8, 9, 10], [11, 12, 13, 14, 15]):
        rows = [0] + chain
        k2, v2 = k.clone(), v.clone()
        for d, r in enumerate(rows):
            k2[root + d], v2[root + d] = k[root + r], v[root + r]
        kc2 = torch.zeros_like(kc); vc2 = torch.zeros_like(vc)
        pack_cache_fast(kc2, vc2, k2, v2, bt[0])
        ref = ops.paged_attention(q[rows].reshape(6, Hq * D).cuda(), kc2.cuda(), vc2.cuda(), bt.cuda(), sl, 0, 1, 6,
                                  Hq, Hkv, D, D ** -0.5, 2048, split).cpu().view(6, Hq, D)
        diffs = [(out[r].float() - ref[d].float()).abs().max().item() for d, r in enumerate(rows)]
        worst[root] = max(worst.get(root, 0.0), max(diffs))
        if root > 1000:
            print(f"root {root} chain {chain}: max diff per row {['%.4f' % x for x in diffs]}")
print("worst diff over roots 1..119, 250..260:", max(v for r, v in worst.items() if r < 1000),
      "at root", max((v, r) for r, v in worst.items() if r < 1000)[1])
print({r: round(v, 4) for r, v in worst.items() if v > 0.004})
