"""Time of one one-shot IPC all-reduce inside a replayed graph, two processes on ONE GPU (no xGMI in
the path: this measures the launch + flag handshake + combine, the floor of the collective):
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 experiments/ipc/bench_ipc_allreduce.py
"""
import ctypes as C
import os, sys
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vllm_omni_draft_amd import _lib, parallel

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
lib = _lib.load()
ipc = parallel.IpcTensorParallelGroup(rank, world, max_message_bytes=4 << 20)
stream = torch.cuda.Stream()
L = 64
for rows, hidden in ((1, 8192), (6, 8192), (6, 4096), (32, 8192), (192, 8192)):
    t = torch.ones(rows * hidden, dtype=torch.bfloat16, device="cuda")
    dist.barrier()
    with torch.cuda.stream(stream):
        for _ in range(3):
            ipc.all_reduce(t)
        torch.cuda.synchronize()
        g = C.c_void_p()
        _lib.check(lib.sd_graph_begin(stream.cuda_stream), "begin")
        for _ in range(L):
            ipc.all_reduce(t)
        _lib.check(lib.sd_graph_end(stream.cuda_stream, C.byref(g)), "end")
        ts = []
        for _ in range(7):
            t.fill_(1.0)
            torch.cuda.synchronize()
            dist.barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            lib.sd_graph_launch(g, stream.cuda_stream)
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / L)
        lib.sd_graph_destroy(g)
    ts.sort()
    if rank == 0:
        print(f"{rows:4d} rows x {hidden} bf16 = {rows * hidden * 2 / 1024:7.1f} KB: {ts[len(ts) // 2]:6.2f} us per "
              f"all-reduce (min {ts[0]:.2f}), timeouts {ipc.timeouts()}", flush=True)
dist.barrier()
ipc.close()
