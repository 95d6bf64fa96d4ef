import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_omni_draft_amd import ops
V, R = 128256, 6
logits = torch.randn(R, V, device="cuda") * 3
base = torch.softmax(logits, dim=1)
for name, kw in (("off", {}), ("top_k 50", dict(k=50)), ("top_p 0.9", dict(p=0.9)), ("top_k 50 + top_p 0.9", dict(k=50, p=0.9))):
    tk = torch.full((1,), kw.get("k", 0), dtype=torch.int32, device="cuda")
    tp = torch.full((1,), kw.get("p", 1.0), dtype=torch.float32, device="cuda")
    mp = torch.zeros(1, device="cuda")
    probs = base.clone()
    ops.filter_probs(probs, tk, tp, mp, rows_per_seq=R)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(20):
        probs.copy_(base)
        e0.record()
        ops.filter_probs(probs, tk, tp, mp, rows_per_seq=R)
        e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    print(f"{name:24s} {ts[len(ts)//2]:7.1f} us for {R} rows of {V}")
