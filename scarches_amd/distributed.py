"""Data-parallel helpers: one process per GPU, torch.distributed over RCCL/xGMI ("nccl" backend on ROCm) or gloo
on CPU for the host-logic tests.

The reference is single-device; cells shard naturally, so the only data-path collective is ONE
all-reduce of the flat gradient buffer per step (<= 12 MB: a single bucket, latency-bound on xGMI), plus scalar loss
sums for logging / early stopping so that every rank takes the same stop / LR decision.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def is_dist() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def allreduce_mean_(flat: torch.Tensor, world: int | None = None) -> torch.Tensor:
    """In-place mean all-reduce of the flat gradient buffer (the whole model is one bucket)."""
    if world is None:
        world = dist.get_world_size()
    if world <= 1:
        return flat
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.mul_(1.0 / world)
    return flat


def allreduce_sum_(t: torch.Tensor) -> torch.Tensor:
    if is_dist():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def shard_range(n: int, rank: int, world: int):
    """Contiguous cell shard [lo, hi) of rank `rank` (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
