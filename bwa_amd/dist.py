"""Multi-GPU plumbing: one process per GPU, reads sharded, index broadcast once over RCCL/xGMI.

The data path has no collective: every rank aligns its own contiguous range of the batch.  The only exchange is the
one-time broadcast of the index buffers (BWT+Occ blocks, SA, pac) from the rank that loaded them; `torch.distributed`
backend "nccl" is RCCL on ROCm, "gloo" works on CPU (used by the world_size-2 tests with the mock-runtime build).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from .api import BwaGpu


def shard_range(n_units: int, rank: int, world: int):
    """Contiguous [lo, hi) of `n_units` (reads, or read pairs for PE) owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(n_units, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class _DeviceBytes:
    """Zero-copy view of a raw HIP device pointer for torch (CUDA array interface v2)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def _as_tensor(ptr: int, nbytes: int, on_gpu: bool) -> torch.Tensor:
    if on_gpu:
        return torch.as_tensor(_DeviceBytes(ptr, nbytes), device="cuda")
    return torch.frombuffer((C.c_uint8 * nbytes).from_address(ptr), dtype=torch.uint8)


def broadcast_index(prefix: str, device: int = 0, src: int = 0, lib_path: str | None = None, chunk: int = 1 << 30) -> BwaGpu:
    """Rank `src` loads <prefix>.* and uploads it; every other rank allocates and receives the buffers by broadcast."""
    rank = dist.get_rank()
    on_gpu = dist.get_backend() == "nccl"
    meta = [None]
    gpu = None
    if rank == src:
        gpu = BwaGpu(prefix, device=device, lib_path=lib_path)
        meta = [gpu.index_meta()]
    dist.broadcast_object_list(meta, src=src)
    if rank != src:
        gpu = BwaGpu.empty(meta[0], device=device, lib_path=lib_path)
    for ptr, nbytes in gpu.index_buffers():
        t = _as_tensor(ptr, nbytes, on_gpu)
        for o in range(0, nbytes, chunk):          # bounded chunks keep RCCL's staging small on multi-GB indices
            dist.broadcast(t[o:o + chunk], src=src)
    if on_gpu:
        torch.cuda.synchronize()
    if rank != src:
        gpu.index_ready()      # derive the prefix tables from the received index
    return gpu


def align_sharded(gpu: BwaGpu, opt, seqs: np.ndarray, off: np.ndarray, pair: bool = False):
    """Align this rank's contiguous share of the batch; returns (lo, hi, counts, regs) in read indices."""
    rank, world = dist.get_rank(), dist.get_world_size()
    n = off.shape[0] - 1
    lo, hi = shard_range(n // 2 if pair else n, rank, world)
    if pair:
        lo, hi = 2 * lo, 2 * hi
    sub_off = off[lo:hi + 1] - off[lo]
    counts, regs = gpu.align(opt, seqs[off[lo]:off[hi]], sub_off)
    return lo, hi, counts, regs
