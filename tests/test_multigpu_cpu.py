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


def _predict_rescaled(page):
    """A predict_fn whose label map does not have its page's shape (Predictor with high_res_output rescales)."""
    lab = _predict(page)
    return np.repeat(np.repeat(lab, 2, 0), 3, 1).astype(np.uint8)


def _predict_bad_dtype(page):
    return _predict(page).astype(np.float32)


def _worker(rank, world, port, q, fn_name="_predict"):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "page-segmentation_amd")]
    from pseg_amd.parallel import predict_pages_sharded, shard_pages
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pages = _pages()
    fn = globals().get(fn_name)
    if fn_name == "_predict_bad_dtype":
        # every rank must raise (nobody is left blocked in a send): the barrier below is reached by both
        try:
            predict_pages_sharded(fn, pages, rank, world)
            raised = False
        except ValueError:
            raised = True
        dist.barrier()
        q.put((rank, raised))
        dist.destroy_process_group()
        return
    if fn_name == "_predict_share_in_one_call":
        # a rank's whole share through ONE call (the route to pseg_predict_batch's page units)
        out = predict_pages_sharded(None, pages, rank, world, batch_fn=lambda ps: [_predict(p_) for p_ in ps])
    else:
        out = predict_pages_sharded(fn, pages, rank, world)
    dist.barrier()
    if rank == 0:
        q.put([o.tolist() for o in out])
    else:
        assert out is None and shard_pages(len(pages), rank, world) == [1, 3]
    dist.destroy_process_group()


def _run_two_ranks(fn_name, n_results=1):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, fn_name)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in range(n_results)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return got


def test_two_rank_page_sharding_matches_single_process(oracle_mod):
    got = _run_two_ranks("_predict")[0]
    want = [_predict(p) for p in _pages()]
    assert len(got) == len(want)
    for g, w_ in zip(got, want):
        assert np.array_equal(np.array(g), w_)


def test_two_rank_page_sharding_with_one_call_per_share(oracle_mod):
    """batch_fn: every rank hands its whole share to one call (on the GPU: Engine.predict_batch, whose same-shape neighbours run as
    page units); the gathered maps are the single-process ones, in page order."""
    got = _run_two_ranks("_predict_share_in_one_call")[0]
    want = [_predict(p) for p in _pages()]
    assert len(got) == len(want)
    for g, w_ in zip(got, want):
        assert np.array_equal(np.array(g), w_)


def test_two_rank_gather_takes_the_senders_shape_and_dtype(oracle_mod):
    """Label maps that do not have their page's shape (a rescaling predict_fn) arrive intact: rank 0 sizes its receive
    buffers from the table the ranks exchange first, not from the page."""
    got = _run_two_ranks("_predict_rescaled")[0]
    want = [_predict_rescaled(p) for p in _pages()]
    for g, w_ in zip(got, want):
        assert np.array_equal(np.array(g, dtype=np.uint8), w_)


def test_two_rank_gather_rejects_a_bad_dtype_on_every_rank(oracle_mod):
    got = dict(_run_two_ranks("_predict_bad_dtype", n_results=2))
    assert got == {0: True, 1: True}


# ---- data-parallel train step: sum of flat gradients over ranks, 1/world at apply -------

def _train_pages():
    from pseg_amd import synth
    return [synth.synth_page(s, 32, 32, 3) for s in range(3)]          # 3 samples on 2 ranks: ragged tail


def _flat(grads, order):
    return np.concatenate([np.asarray(grads[k], np.float32).ravel() for k in order])


def _dp_worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "page-segmentation_amd")]
    import torch
    import oracle
    from oracle import train_ref
    from pseg_amd.parallel import dp_train_epoch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    Wt = oracle.init_weights("fcn", 3, seed=2, gain=1.0, bias_scale=0.05)
    order = list(Wt)
    opt = train_ref.KerasAdam(lr=1e-3, clipnorm=1.0)
    pages = _train_pages()
    state = {}

    def fb(k):
        img, _, mask = pages[k]
        loss, _, _, _, grads = train_ref.fcn_loss_and_grads("fcn", Wt, img, mask)
        state["g"] = torch.from_numpy(_flat(grads, order).copy())
        return loss

    def apply(scale):
        g = state["g"].numpy() * np.float32(scale)
        off, grads = 0, {}
        for k in order:
            n = Wt[k].size
            grads[k] = g[off:off + n].reshape(Wt[k].shape)
            off += n
        opt.apply(Wt, grads)

    dp_train_epoch(len(pages), rank, world, fb, lambda: state["g"], apply)
    dist.barrier()
    q.put((rank, {k: v.tolist() for k, v in Wt.items()}))
    dist.destroy_process_group()


def test_two_rank_dp_train_matches_gradient_averaging(oracle_mod):
    import oracle
    from oracle import train_ref
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single process: step 1 averages pages 0 and 1, step 2 takes page 2 alone
    Wt = oracle.init_weights("fcn", 3, seed=2, gain=1.0, bias_scale=0.05)
    opt = train_ref.KerasAdam(lr=1e-3, clipnorm=1.0)
    pages = _train_pages()
    for group in ([0, 1], [2]):
        gs = [train_ref.fcn_loss_and_grads("fcn", Wt, pages[k][0], pages[k][2])[4] for k in group]
        avg = {k: sum(np.asarray(g[k], np.float32) for g in gs) * np.float32(1.0 / len(group)) for k in Wt}
        opt.apply(Wt, avg)
    for r in (0, 1):                                   # replicas stay identical and match
        for k in Wt:
            np.testing.assert_allclose(np.array(got[r][k], np.float32), Wt[k], rtol=1e-5, atol=1e-7)
    for k in Wt:
        assert np.array_equal(np.array(got[0][k], np.float32), np.array(got[1][k], np.float32))
