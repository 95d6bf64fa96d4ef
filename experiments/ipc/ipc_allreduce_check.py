"""Two processes on one GPU: the library's one-shot IPC all-reduce against a gloo reference.
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 experiments/ipc/ipc_allreduce_check.py
"""
import os, sys, time
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vllm_omni_draft_amd import parallel

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
ipc = parallel.IpcTensorParallelGroup(rank, world, max_message_bytes=1 << 20)
dist.barrier()
g = torch.Generator(device="cuda").manual_seed(100 + rank)
for n in (8, 64, 4096, 4104, 400000, 524288):
    mine = (torch.randn(n, generator=g, device="cuda") * 3).to(torch.bfloat16)
    host = [torch.empty(n, dtype=torch.bfloat16) for _ in range(world)]
    dist.all_gather(host, mine.cpu())
    want = sum(h.float() for h in host).to(torch.bfloat16)
    for rep in range(3):
        t = mine.clone()
        t0 = time.perf_counter()
        ipc.all_reduce(t)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"rank {rank} n={n} rep {rep}: equal {torch.equal(t.cpu(), want)} in {dt * 1e3:.2f} ms, "
              f"timeouts {ipc.timeouts()}", flush=True)
dist.barrier()
ipc.close()
