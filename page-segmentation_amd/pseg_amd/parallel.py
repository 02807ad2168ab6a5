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


def predict_pages_sharded(predict_fn: Optional[Callable[[np.ndarray], np.ndarray]], pages: Sequence[np.ndarray],
                          rank: int = 0, world: int = 1, gather: bool = True,
                          batch_fn: Optional[Callable[[List[np.ndarray]], List[np.ndarray]]] = None) -> Optional[List[np.ndarray]]:
    """Every rank predicts its own pages with `predict_fn(page) -> label map` -- or, given `batch_fn(list of pages) -> list of
    label maps` (e.g. `lambda ps: engine.predict_batch(ps, dtype=np.uint8)`), its whole share in ONE call, so that the share
    travels through pseg_predict_batch's page units (same-shape neighbours computed together); with gather=True
    rank 0 returns all label maps in page order (other ranks return None).  Pages may differ
    in size (ragged) and a rank may own no page at all.  Label maps travel as plain tensors, one point-to-point
    message per page (shapes and dtypes are exchanged first in one small integer table; nothing is pickled)."""
    mine = shard_pages(len(pages), rank, world)
    if batch_fn is not None:
        maps = batch_fn([pages[i] for i in mine]) if mine else []
        if len(maps) != len(mine):
            raise ValueError("batch_fn returned %d label maps for %d pages" % (len(maps), len(mine)))
        local = [(i, np.ascontiguousarray(m)) for i, m in zip(mine, maps)]
    else:
        local = [(i, np.ascontiguousarray(predict_fn(pages[i]))) for i in mine]
    if world == 1:
        return [lab for _, lab in local]
    if not gather:
        return None
    import torch
    import torch.distributed as dist
    on_gpu = dist.get_backend() == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
    # Every rank publishes (height, width, dtype code) of each of its label maps BEFORE any map travels: rank 0 sizes
    # its receive buffers from what the sender really holds (a predict_fn may rescale its output, e.g. high_res_output),
    # and an unsupported dtype makes EVERY rank raise together instead of leaving the senders blocked in dist.send.
    dcode = {np.dtype(np.uint8): 1, np.dtype(np.int64): 2, np.dtype(np.int32): 3}
    meta = torch.zeros((len(pages), 3), dtype=torch.int64, device=dev)
    for i, lab in local:
        if lab.ndim != 2:
            meta[i] = torch.tensor([0, 0, -1], dtype=torch.int64)
        else:
            meta[i] = torch.tensor([lab.shape[0], lab.shape[1], dcode.get(lab.dtype, -1)], dtype=torch.int64)
    dist.all_reduce(meta, op=dist.ReduceOp.SUM)      # disjoint rows per rank: the sum is the table
    meta_h = meta.cpu().numpy()
    bad = [i for i in range(len(pages)) if meta_h[i, 2] not in (1, 2, 3)]
    if bad:
        raise ValueError("page(s) %r: label maps must be 2-D uint8 / int32 / int64 arrays (rank %d sends page %d)"
                         % (bad, bad[0] % world, bad[0]))
    if rank != 0:
        for _, lab in local:
            dist.send(torch.from_numpy(lab).to(dev), dst=0)
        return None
    out: List[Optional[np.ndarray]] = [None] * len(pages)
    for i, lab in local:
        out[i] = lab
    tdt = {1: torch.uint8, 2: torch.int64, 3: torch.int32}
    for i in range(len(pages)):                      # page order = interleaved rank order: every sender's queue is drained in its send order
        r = i % world
        if r == 0:
            continue
        buf = torch.empty((int(meta_h[i, 0]), int(meta_h[i, 1])), dtype=tdt[int(meta_h[i, 2])], device=dev)
        dist.recv(buf, src=r)
        out[i] = buf.cpu().numpy()
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


def allreduce_flat(flat, world, engine_stream=None):
    """Sum the flat gradient vector over the ranks in place (RCCL on device tensors, gloo on CPU
    tensors in the tests).  The 1/world factor is applied by train_apply(grad_scale=1/world).

    The engine writes / reads the buffer on its own HIP stream.  With `engine_stream` (the raw hipStream_t from
    Engine.stream()) the collective is ENQUEUED in stream order -- RCCL's stream waits for the backward kernels on the
    engine stream, the engine stream then waits for the collective, the host never blocks -- so the clip + Adam
    kernels of train_apply queue up behind it.  (Bucketed overlap with the tail of backward is not worth having: the
    whole gradient is 2.7 MB, ~25 us over xGMI against a 38 ms step.)  Without it the device is synchronised on both sides."""
    if world <= 1:
        return flat
    import torch
    import torch.distributed as dist
    if not flat.is_cuda:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        return flat
    if dist.get_backend() != "nccl":
        # gloo rehearsal on a GPU box: stage through the host
        torch.cuda.synchronize(flat.device)
        host = flat.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        flat.copy_(host)
        torch.cuda.synchronize(flat.device)
        return flat
    if engine_stream:
        ext = torch.cuda.ExternalStream(int(engine_stream), device=flat.device)
        with torch.cuda.stream(ext):
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)     # blocking form = the CURRENT (engine) stream waits, not the host
        return flat
    torch.cuda.synchronize(flat.device)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    torch.cuda.synchronize(flat.device)
    return flat


def allreduce_gradients(engine, world):
    """the only collective of the build -- one RCCL all-reduce(sum) of the flat fp32
    gradient (673 013 + metric slots for fcn_skip C=3, 2.7 MB) per train step."""
    if world <= 1:
        return
    allreduce_flat(grad_tensor(engine), world, engine_stream=engine.stream())


def dp_train_epoch(n_samples, rank, world, forward_backward, flat_gradient, apply, engine_stream=None):
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
            if engine_stream and g.is_cuda:
                import torch
                with torch.cuda.stream(torch.cuda.ExternalStream(int(engine_stream), device=g.device)):
                    g.zero_()           # behind the backward kernels that wrote it, on the engine's stream
            else:
                g.zero_()
        allreduce_flat(g, world, engine_stream=engine_stream)
        contributors = min(world, n_samples - st * world)
        apply(1.0 / contributors)
        if live:
            rows.append(m)
    return rows
