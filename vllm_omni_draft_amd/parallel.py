"""Multi-GPU layout: one process per GPU (torch.distributed; backend "nccl" is
RCCL on ROCm, "gloo" on CPU for tests).

Data-parallel replicas: requests are dealt round-robin to N single-GPU engines,
each with its own target, draft and KV cache.  Nothing is exchanged on the data
path; only throughput accounting crosses ranks.  (The reference runs a single
uni-process executor, /root/reference/src/qwen_image_gen/executor.py:11.)

Tensor-parallel target (a 70B target over the 8 GPUs of a node): every rank holds
its shard of the target (weights.shard_state_dict) and a whole draft; the native
forward sums the two row-parallel projections of each layer over the ranks with
RCCL (TensorParallelGroup).  Ranks run the same step on the same inputs, so
proposals, acceptance and state are identical everywhere with no further exchange.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Callable, List, Optional, Sequence, Tuple, TypeVar

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


class TensorParallelGroup:
    """An RCCL communicator owned by the native library (sd_tp_comm_create).

    broadcast_id(id_bytes_or_None) -> bytes spreads rank 0's 128-byte id to every rank;
    by default it goes through torch.distributed (any backend)."""

    def __init__(self, rank: int, world_size: int,
                 broadcast_id: Optional[Callable[[Optional[bytes]], bytes]] = None):
        from . import _lib
        lib = _lib.load()
        self.rank, self.world_size = rank, world_size
        n = 128  # SD_TP_ID_BYTES
        mine = None
        if rank == 0:
            buf = (C.c_ubyte * n)()
            _lib.check(lib.sd_tp_unique_id(buf), "sd_tp_unique_id")
            mine = bytes(buf)
        uid = (broadcast_id or _broadcast_bytes)(mine) if world_size > 1 else mine
        if uid is None or len(uid) != n:
            raise RuntimeError("tensor-parallel id exchange failed")
        self._comm = lib.sd_tp_comm_create((C.c_ubyte * n).from_buffer_copy(uid), rank, world_size)
        if not self._comm:
            raise _lib.SdError(f"sd_tp_comm_create failed: {_lib.last_error()}")
        self.allreduce_ptr = C.cast(lib.sd_tp_allreduce, C.c_void_p).value
        self.ctx_ptr = self._comm

    def all_reduce(self, t: torch.Tensor) -> None:
        """In-place sum of a bf16 device tensor over the group (current stream)."""
        from . import _lib
        if t.dtype != torch.bfloat16 or not t.is_cuda or not t.is_contiguous():
            raise _lib.SdError("all_reduce: contiguous bf16 device tensor expected")
        _lib.check(_lib.load().sd_tp_allreduce(t.data_ptr(), t.numel(), self._comm,
                                               torch.cuda.current_stream().cuda_stream),
                   "sd_tp_allreduce")

    def all_gather(self, send: torch.Tensor, recv: torch.Tensor) -> None:
        """recv = the ranks' `send` buffers one after the other, in rank order (current stream)."""
        from . import _lib
        nbytes = send.numel() * send.element_size()
        if not (send.is_cuda and recv.is_cuda and send.is_contiguous() and recv.is_contiguous()) \
                or recv.numel() * recv.element_size() != nbytes * self.world_size:
            raise _lib.SdError("all_gather: contiguous device tensors, recv = world_size x send")
        _lib.check(_lib.load().sd_tp_allgather(send.data_ptr(), recv.data_ptr(), nbytes, self._comm,
                                               torch.cuda.current_stream().cuda_stream),
                   "sd_tp_allgather")

    def close(self) -> None:
        if getattr(self, "_comm", None):
            from . import _lib
            _lib.load().sd_tp_comm_destroy(self._comm)
            self._comm = None


class IpcTensorParallelGroup:
    """One-shot all-reduce / all-gather over peer-mapped device memory (sd_ipc_*): for the ranks
    of one node, one process per GPU.  Every rank creates an exchange buffer, the 64-byte IPC
    handles travel through `exchange` (default: torch.distributed.all_gather_object on the
    default group, any backend), every rank maps its peers' buffers.  Same surface as
    TensorParallelGroup (rank, world_size, allreduce_ptr, ctx_ptr, all_reduce, all_gather), so
    an engine takes either.  max_message_bytes bounds rows x hidden x 2 of a pass (a
    256-row prefill pass at hidden 8192 is 4 MiB)."""

    def __init__(self, rank: int, world_size: int, max_message_bytes: int = 4 << 20,
                 exchange: Optional[Callable[[bytes], List[bytes]]] = None):
        from . import _lib
        lib = _lib.load()
        self.rank, self.world_size = rank, world_size
        self._lib = lib
        self._own = C.c_void_p()
        self._peers: List[Optional[int]] = [None] * world_size
        self._comm = None
        handle = (C.c_ubyte * 64)()
        nbytes = lib.sd_ipc_buffer_bytes(max_message_bytes)
        _lib.check(lib.sd_ipc_buffer_create(nbytes, C.byref(self._own), handle), "sd_ipc_buffer_create")
        handles = (exchange or _all_gather_bytes)(bytes(handle)) if world_size > 1 else [bytes(handle)]
        if len(handles) != world_size or any(len(h) != 64 for h in handles):
            raise RuntimeError("IPC handle exchange failed")
        bufs = (C.c_void_p * world_size)()
        for q in range(world_size):
            if q == rank:
                bufs[q] = self._own.value
                continue
            p = C.c_void_p()
            _lib.check(lib.sd_ipc_buffer_open((C.c_ubyte * 64).from_buffer_copy(handles[q]), C.byref(p)),
                       "sd_ipc_buffer_open")
            self._peers[q] = p.value
            bufs[q] = p.value
        self._comm = lib.sd_ipc_comm_create(rank, world_size, bufs, max_message_bytes)
        if not self._comm:
            raise _lib.SdError(f"sd_ipc_comm_create failed: {_lib.last_error()}")
        self.allreduce_ptr = C.cast(lib.sd_ipc_allreduce, C.c_void_p).value
        self.ctx_ptr = self._comm

    def all_reduce(self, t: torch.Tensor) -> None:
        """In-place sum of a bf16 device tensor over the group (current stream)."""
        from . import _lib
        if t.dtype != torch.bfloat16 or not t.is_cuda or not t.is_contiguous():
            raise _lib.SdError("all_reduce: contiguous bf16 device tensor expected")
        _lib.check(self._lib.sd_ipc_allreduce(t.data_ptr(), t.numel(), self._comm,
                                              torch.cuda.current_stream().cuda_stream), "sd_ipc_allreduce")

    def all_gather(self, send: torch.Tensor, recv: torch.Tensor) -> None:
        """recv = the ranks' `send` buffers one after the other, in rank order (current stream)."""
        from . import _lib
        nbytes = send.numel() * send.element_size()
        if not (send.is_cuda and recv.is_cuda and send.is_contiguous() and recv.is_contiguous()) \
                or recv.numel() * recv.element_size() != nbytes * self.world_size:
            raise _lib.SdError("all_gather: contiguous device tensors, recv = world_size x send")
        _lib.check(self._lib.sd_ipc_allgather(send.data_ptr(), recv.data_ptr(), nbytes, self._comm,
                                              torch.cuda.current_stream().cuda_stream), "sd_ipc_allgather")

    def set_timeout_ms(self, milliseconds: int) -> None:
        """Bound of a wait for a peer in the calls enqueued from now on (0: the default, ~20 s).
        Freshly started processes may lag each other by more than that on the host side (cold
        page cache, first launches): run a warm-up pass under a generous bound, then restore it
        before anything is captured into a graph."""
        from . import _lib
        _lib.check(self._lib.sd_ipc_comm_set_timeout_ms(self._comm, int(milliseconds)),
                   "sd_ipc_comm_set_timeout_ms")

    def warm_up(self, nbytes: int) -> int:
        """One all-reduce of `nbytes` (zeros) on the current stream, waited for: the first
        exchange of a group maps the peers' pages and may find a peer seconds behind (still
        loading weights); doing it here, before the first timed or served pass, keeps that out of
        the passes.  Returns the waits that gave up during it (0: every peer arrived)."""
        before = self.timeouts()
        t = torch.zeros(max(nbytes // 2, 8), dtype=torch.bfloat16, device="cuda")
        self.all_reduce(t)
        torch.cuda.current_stream().synchronize()
        return self.timeouts() - before

    def timeouts(self) -> int:
        """Waits that gave up since the group was made (synchronises); 0 in a healthy run."""
        from . import _lib
        n = C.c_int(0)
        _lib.check(self._lib.sd_ipc_comm_errors(self._comm, C.byref(n)), "sd_ipc_comm_errors")
        return n.value

    def close(self) -> None:
        """Unmaps the peers' buffers and frees this rank's.  Every rank must have finished its
        last collective (synchronise the ranks first): a peer may still be reading."""
        if self._comm:
            self._lib.sd_ipc_comm_destroy(self._comm)
            self._comm = None
        for q, p in enumerate(self._peers):
            if p:
                self._lib.sd_ipc_buffer_close(p)
                self._peers[q] = None
        if self._own.value:
            self._lib.sd_ipc_buffer_destroy(self._own)
            self._own = C.c_void_p()


def _all_gather_bytes(payload: bytes) -> List[bytes]:
    out: List = [None] * dist.get_world_size()
    dist.all_gather_object(out, payload)
    return out


def _broadcast_bytes(payload: Optional[bytes]) -> bytes:
    """Rank 0's payload on every rank of the default process group."""
    box = [payload]
    dist.broadcast_object_list(box, src=0)
    return box[0]


class CallbackAllReduce:
    """A tensor-parallel 'group' whose all-reduce is a host callback
    fn(ptr, n_bf16, stream) -> int: for tests and for hosts with their own collective."""

    def __init__(self, rank: int, fn: Callable[[int, int, int], int], world_size: int = 2,
                 gather: Optional[Callable[[torch.Tensor, torch.Tensor], None]] = None):
        from . import _lib
        self.rank, self.world_size, self._gather = rank, world_size, gather
        self._cb = _lib.ALLREDUCE_FN(lambda buf, n, ctx, stream: fn(buf, n, stream or 0))
        self.allreduce_ptr = C.cast(self._cb, C.c_void_p).value
        self.ctx_ptr = None

    def all_gather(self, send: torch.Tensor, recv: torch.Tensor) -> None:
        if self._gather is None:
            raise RuntimeError("this group was built without an all-gather")
        self._gather(send, recv)
