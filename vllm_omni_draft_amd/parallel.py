"""Multi-GPU layout: one process per GPU (torch.distributed; backend "nccl" is
RCCL on ROCm, "gloo" on CPU for tests).

Data-parallel replicas: requests are dealt round-robin to N single-GPU engines,
each with its own target, draft and KV cache.  Nothing is exchanged on the data
path; only throughput accounting crosses ranks.  (The reference runs a single
uni-process executor, /root/reference/src/qwen_image_gen/executor.py:11.)
"""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple, TypeVar

import torch
import torch.distributed as dist

T = TypeVar("T")


def init_from_env(backend: str = "nccl") -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun environment; initialises
    the default process group when world_size > 1."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def shard_requests(requests: Sequence[T], rank: int, world: int) -> List[T]:
    """Round-robin assignment: request i runs on rank i % world."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside 0..{world - 1}")
    return [r for i, r in enumerate(requests) if i % world == rank]


def owner_of(request_index: int, world: int) -> int:
    return request_index % world


def aggregate_throughput(tokens: int, seconds: float, device=None) -> Tuple[float, float]:
    """Whole-job (total tokens, wall seconds): tokens summed over ranks, time the
    maximum over ranks.  Single process: returns the inputs."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(tokens), float(seconds)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    n = torch.tensor([float(tokens)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(n, op=dist.ReduceOp.SUM)
    return float(n.item()), float(t.item())


def gather_streams(local: List[List[int]], world: int) -> List[List[List[int]]]:
    """Token streams of every rank, on every rank (for result collection)."""
    if not dist.is_initialized() or world == 1:
        return [local]
    out: List = [None] * world
    dist.all_gather_object(out, local)
    return out
