"""Page-parallel predict across the GPUs of one node: independent pages, static
round-robin page -> rank (page i -> rank i mod world), weights replicated, NO data-path
collective.  torch.distributed (RCCL on GPUs, gloo in the CPU tests) is used only to hand the
label maps to rank 0 and for barriers/timing; a single huge page is not split.
"""
from typing import Callable, List, Optional, Sequence

import numpy as np


def shard_pages(n_pages: int, rank: int, world: int) -> List[int]:
    """Indices of the pages rank `rank` owns (round-robin, as SURVEY.md 8d config 3)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank %d / world %d" % (rank, world))
    return list(range(rank, n_pages, world))


def predict_pages_sharded(predict_fn: Callable[[np.ndarray], np.ndarray], pages: Sequence[np.ndarray],
                          rank: int = 0, world: int = 1, gather: bool = True) -> Optional[List[np.ndarray]]:
    """Every rank predicts its own pages with `predict_fn(page) -> label map`; with gather=True
    rank 0 returns all label maps in page order (other ranks return None).  Pages may differ
    in size (ragged) and a rank may own no page at all."""
    mine = shard_pages(len(pages), rank, world)
    local = [(i, np.ascontiguousarray(predict_fn(pages[i]))) for i in mine]
    if world == 1:
        return [lab for _, lab in local]
    if not gather:
        return None
    import torch.distributed as dist
    bucket = [None] * world if rank == 0 else None
    dist.gather_object(local, bucket, dst=0)
    if rank != 0:
        return None
    out: List[Optional[np.ndarray]] = [None] * len(pages)
    for part in bucket:
        for i, lab in part:
            out[i] = lab
    return out  # type: ignore[return-value]


class _DevBuf:
    """__cuda_array_interface__ view of a raw device pointer (float32 vector)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2}


def grad_tensor(engine):
    """Zero-copy torch view of the engine's flat gradient buffer (all parameters + the metric
    accumulators).  Data-parallel training = one all-reduce on this tensor, then
    engine.train_apply(lr, grad_scale=1/world) on every rank."""
    import torch
    ptr, n = engine.grad_buffer()
    return torch.as_tensor(_DevBuf(ptr, n), device="cuda:%d" % engine.device)


def allreduce_flat(flat, world):
    """Sum the flat gradient vector over the ranks in place (RCCL on device tensors, gloo on CPU
    tensors in the tests).  The 1/world factor is applied by train_apply(grad_scale=1/world)."""
    if world <= 1:
        return flat
    import torch
    import torch.distributed as dist
    if flat.is_cuda:                       # the engine writes / reads the buffer on its own stream
        torch.cuda.synchronize(flat.device)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if flat.is_cuda:
        torch.cuda.synchronize(flat.device)
    return flat


def allreduce_gradients(engine, world):
    """the only collective of the build -- one RCCL all-reduce(sum) of the flat fp32
    gradient (673 013 + metric slots for fcn_skip C=3, 2.7 MB) per train step."""
    if world <= 1:
        return
    allreduce_flat(grad_tensor(engine), world)


def dp_train_epoch(n_samples, rank, world, forward_backward, flat_gradient, apply):
    """One data-parallel pass: rank r takes samples r, r+world, ... (ranks that run out repeat
    their last sample with zero weight so that every rank enters every all-reduce), the flat
    gradients are summed over the ranks and applied with scale 1/contributors."""
    steps = (n_samples + world - 1) // world
    rows = []
    for st in range(steps):
        k = st * world + rank
        live = k < n_samples
        m = forward_backward(k if live else n_samples - 1)
        g = flat_gradient()
        if not live:
            g.zero_()
        allreduce_flat(g, world)
        contributors = min(world, n_samples - st * world)
        apply(1.0 / contributors)
        if live:
            rows.append(m)
    return rows
