"""The few device-runtime calls the host code makes (streams, synchronisation, pinned memory),
for the GPU library and for the CPU build of the ABI (SD_AMD_BACKEND=cpu: no streams, every
call is synchronous, host memory needs no pinning)."""
from __future__ import annotations

import contextlib

import torch

from . import _lib


def on_gpu() -> bool:
    return _lib.backend() == "gpu"


def device(index: int = 0) -> torch.device:
    return torch.device("cuda", index) if on_gpu() else torch.device("cpu")


def new_stream(dev):
    return torch.cuda.Stream(device=dev) if on_gpu() else None


def stream_ctx(stream):
    return torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()


def synchronize(dev=None) -> None:
    if on_gpu():
        torch.cuda.synchronize(dev)


def sync_current_stream() -> None:
    if on_gpu():
        torch.cuda.current_stream().synchronize()


def pin(t: torch.Tensor) -> torch.Tensor:
    return t.pin_memory() if on_gpu() else t
