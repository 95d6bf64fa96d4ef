"""One rank's share of the tensor-parallel configuration (Llama-3-70B target over 8 ranks, k = 4,
Llama-3-8B draft) on ONE GPU: the rank's weight shards with random values, an RCCL group of one
rank (the collectives are launched but move nothing), the step replayed as a graph.  It measures
what a rank computes per step, without the xGMI exchange: a lower bound of the 8-GPU step time,
and the comparison of a replicated draft with a draft sharded like the target.
    python experiments/bench_tp_share.py [B] [prompt_len] [tp]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_omni_draft_amd import config, parallel, weights  # noqa: E402
from vllm_omni_draft_amd.engine import EngineConfig, SpecEngine  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
P = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
TP = int(sys.argv[3]) if len(sys.argv) > 3 else 8
k, steps, warm, alpha = 4, 24, 4, 0.8
full_t, full_d = config.preset("llama-3-70b"), config.preset("llama-3-8b")
tcfg = full_t.tp_shard(TP, vocab_parallel=True)
tw = weights.random_init(tcfg, seed=0)


class ShareGroup(parallel.TensorParallelGroup):
    """An RCCL group of one rank that stands in for rank 0 of `pretend` ranks: all-reduces are
    launched on the real communicator (and move nothing), the head's all-gather copies the
    rank's own slice into every slot (tokens come from slice 0 only: arbitrary but fixed)."""

    def __init__(self, pretend):
        super().__init__(0, 1)
        self.world_size = pretend

    def all_gather(self, send, recv):
        recv.view(self.world_size, -1).copy_(send.reshape(1, -1).expand(self.world_size, -1))


group = ShareGroup(TP)
g = torch.Generator().manual_seed(7)
prompts = [torch.randint(0, full_t.vocab, (P,), generator=g).tolist() for _ in range(B)]
n_new = (steps + warm) * (k + 1) + 8
max_seq = P + n_new + 16
rows = "wide" if B * (k + 1) > 16 else "narrow"


def run(eng, n_steps, n_warm):
    outs = [[eng.add_request(i, p)] for i, p in enumerate(prompts)]
    for _ in range(n_warm):
        for o, new in zip(outs, eng.step()):
            o.extend(new)
    torch.cuda.synchronize()
    n0 = sum(len(o) for o in outs)
    t0 = time.perf_counter()
    for _ in range(n_steps):
        for o, new in zip(outs, eng.step()):
            o.extend(new)
    torch.cuda.synchronize()
    return outs, sum(len(o) for o in outs) - n0, time.perf_counter() - t0


def engine(kk, dw=None, draft_tp=False):
    return SpecEngine(tw, dw, EngineConfig(k=kk, max_batch=B, max_seq_len=max_seq, rows=rows,
                                           gen_capacity=n_new + 64), tp_group=group, draft_tp=draft_tp)


ar_out, ar_n, ar_dt = run(engine(0), n_new - 8 - warm, warm)
print(f"rank share of 70B/TP{TP}, B={B}: AR {ar_dt / (n_new - 8 - warm) * 1e3:.3f} ms/step, "
      f"{ar_n / ar_dt:.1f} tok/s", flush=True)
truth = torch.zeros(B, max_seq, dtype=torch.int32)
for b in range(B):
    truth[b, :P] = torch.tensor(prompts[b])
    truth[b, P:P + len(ar_out[b])] = torch.tensor(ar_out[b])
for name, dcfg, draft_tp in (("8B draft replicated", full_d, False),
                             (f"8B draft sharded TP{TP}, head whole", full_d.tp_shard(TP), True),
                             (f"8B draft sharded TP{TP}, head split",
                              full_d.tp_shard(TP, vocab_parallel=True), True)):
    dw = weights.random_init(dcfg, seed=1)
    eng = engine(k, dw, draft_tp)
    eng.set_synthetic_acceptance(truth.cuda(), alpha, seed=3)
    out, n, dt = run(eng, steps, warm)
    exact = all(o[:min(len(o), len(a))] == a[:min(len(o), len(a))] for o, a in zip(out, ar_out))
    print(f"  {name}: {dt / steps * 1e3:.3f} ms/step, {n / dt:.1f} tok/s, "
          f"{n / steps / B:.2f} tokens/step/seq, speedup {n / dt / (ar_n / ar_dt):.3f}, "
          f"spec == AR: {exact}", flush=True)
    del eng, dw
    torch.cuda.empty_cache()
