"""N > 1 path on CPU: two processes over gloo shard the pages round-robin, predict their own
pages (here with the CPU oracle standing in for the engine) and rank 0 gathers the label maps:
identical to the single-process result, no data-path collective."""
import os
import socket

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _pages():
    rng = np.random.default_rng(0)
    shapes = [(32, 32), (40, 24), (16, 48), (33, 20), (24, 24)]          # ragged, 5 pages on 2 ranks
    return [rng.integers(0, 256, s, dtype=np.uint8) for s in shapes]


def _predict(page):
    import oracle
    Wt = oracle.init_weights("fcn", 3, seed=1, gain=1.5, bias_scale=0.05)
    return np.argmax(oracle.forward("fcn", Wt, page), -1)


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "page-segmentation_amd")]
    from pseg_amd.parallel import predict_pages_sharded, shard_pages
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pages = _pages()
    out = predict_pages_sharded(_predict, pages, rank, world)
    dist.barrier()
    if rank == 0:
        q.put([o.tolist() for o in out])
    else:
        assert out is None and shard_pages(len(pages), rank, world) == [1, 3]
    dist.destroy_process_group()


def test_two_rank_page_sharding_matches_single_process(oracle_mod):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = [_predict(p) for p in _pages()]
    assert len(got) == len(want)
    for g, w_ in zip(got, want):
        assert np.array_equal(np.array(g), w_)
