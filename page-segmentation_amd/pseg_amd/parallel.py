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


def allreduce_gradients(engine, world):
    """the only collective of the build -- one RCCL all-reduce(sum) of the flat fp32
    gradient (673 013 + metric slots for fcn_skip C=3, 2.7 MB) per train step."""
    if world <= 1:
        return
    import torch
    import torch.distributed as dist
    g = grad_tensor(engine)
    torch.cuda.synchronize(g.device)
    dist.all_reduce(g, op=dist.ReduceOp.SUM)
    torch.cuda.synchronize(g.device)
