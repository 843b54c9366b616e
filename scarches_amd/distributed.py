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


def _host_staged(t: torch.Tensor) -> bool:
    """gloo is the CPU backend (tests, one-GPU rehearsals of the multi-rank path): device tensors go through the host
    explicitly instead of through ProcessGroupGloo's own device path (worker-thread copies on private streams), which
    hung intermittently with two ranks sharing one GPU.  RCCL ("nccl") takes device tensors as they are."""
    return t.is_cuda and dist.get_backend() == "gloo"


def all_reduce_sum_(t: torch.Tensor) -> torch.Tensor:
    """In-place sum over ranks on the tensor's current stream semantics (RCCL) or staged through the host (gloo)."""
    if _host_staged(t):
        h = t.detach().cpu()            # synchronises with the work queued on the current stream
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def broadcast_(t: torch.Tensor, src: int = 0) -> torch.Tensor:
    """In-place broadcast from rank `src` (initial parameter / buffer synchronisation of the replicas)."""
    if not is_dist():
        return t
    if _host_staged(t):
        h = t.detach().cpu()
        dist.broadcast(h, src)
        t.copy_(h)
    else:
        dist.broadcast(t, src)
    return t


def all_gather_rows_(out: torch.Tensor, local: torch.Tensor) -> torch.Tensor:
    """out[world * n, ...] <- the ranks' equally sized `local[n, ...]` blocks in rank order."""
    if _host_staged(local):
        ho, hl = torch.empty(out.shape, dtype=out.dtype), local.detach().cpu()
        dist.all_gather_into_tensor(ho, hl)
        out.copy_(ho)
    else:
        dist.all_gather_into_tensor(out, local)
    return out


def allreduce_mean_(flat: torch.Tensor, world: int | None = None, weight: float | None = None) -> torch.Tensor:
    """In-place all-reduce of the flat gradient buffer (the whole model is one bucket).

    Every rank holds the gradient of the MEAN loss over its local cells.  With equal shards the global-batch
    gradient is the plain mean over ranks; with ragged shards (last partial minibatch) pass
    weight = n_local / n_global and the weighted sum is exact."""
    if world is None:
        world = dist.get_world_size()
    if world <= 1:
        return flat
    if weight is not None:
        flat.mul_(float(weight))
        return all_reduce_sum_(flat)
    all_reduce_sum_(flat)
    flat.mul_(1.0 / world)
    return flat


def local_slice(ids: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """This rank's contiguous slice of a GLOBAL minibatch of row ids (every rank draws the same global sample from
    the same seed, so the slices are disjoint and cover the minibatch: stratified sampling with replacement keeps
    its single-device semantics)."""
    lo, hi = shard_range(ids.numel(), rank, world)
    return ids[lo:hi]


def allreduce_sum_(t: torch.Tensor) -> torch.Tensor:
    if is_dist():
        all_reduce_sum_(t)
    return t


def shard_range(n: int, rank: int, world: int):
    """Contiguous cell shard [lo, hi) of rank `rank` (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
