"""BASELINE.json configs[2] and configs[3] at their real sizes.

configs[2] ("batch of 256 independent 2048x1536 pages sharded page-parallel across 8 GPUs"): one rank's share -- 32
full-size pages through pseg_predict_batch (lib/predictor.py:27-30's page loop), uint8 and int64 label maps, equal to
the page-by-page device entry.
configs[3] ("train loop, 3-class, synthetic masks, DP grad all-reduce"): one full-size 2048x1536 train step against
torch-CPU autograd of the same graph (oracle/train_ref.py; bar from north_star: loss within 1e-4 relative), further
steps with determinism / decreasing-loss properties, and the real Network.train_dataset(rank, world) data-parallel
path with two ranks (gloo rendezvous, both ranks on this box's GPU) against a single-process gradient average
(lib/network.py:235-241 is batch-of-one fit; the DP reference is the averaged gradient)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config2_one_rank_share_32_full_size_pages(gpu):
    import torch
    from pseg_amd import synth
    H, W, C, N = 2048, 1536, 3, 32
    eng = gpu.Engine("fcn_skip", C, mode=gpu.MODE_BF16)
    eng.set_weights(synth.glorot_weights(eng.weight_specs(), seed=42, gain=1.5, bias_scale=0.05))
    base = [synth.synth_page(i, H, W, C)[0] for i in range(4)]
    # 32 distinct pages from 4 synthetic ones (flips / transposed blocks keep the statistics, cost no generator time)
    pages = []
    for i in range(N):
        p = base[i % 4]
        k = i // 4
        p = p[::-1] if k & 1 else p
        p = p[:, ::-1] if k & 2 else p
        p = np.roll(p, 37 * k, axis=1) if k & 4 else p
        pages.append(np.ascontiguousarray(p))
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev).cuda_stream
    want = []
    lab = torch.empty((H, W), dtype=torch.uint8, device=dev)
    for p in pages:
        d = torch.from_numpy(p).to(dev)
        eng.predict_device(d.data_ptr(), H, W, d_labels_u8=lab.data_ptr(), stream=st)
        torch.cuda.synchronize()
        want.append(lab.cpu().numpy().copy())
    assert len({w.tobytes() for w in want}) == N                       # the pages really differ
    got8 = eng.predict_batch(pages, dtype=np.uint8)
    assert all(np.array_equal(g, w) for g, w in zip(got8, want))
    got64 = eng.predict_batch(pages, dtype=np.int64)
    assert all(g.dtype == np.int64 and np.array_equal(g, w) for g, w in zip(got64, want))
    # pinned pages / label maps (the DMA goes straight to the caller's buffers) give the same maps
    ppages = [gpu.pinned_copy(p) for p in pages[:8]]
    pouts = [gpu.pinned_empty((H, W), np.uint8) for _ in range(8)]
    eng.predict_batch(ppages, dtype=np.uint8, out=pouts)
    assert all(np.array_equal(g, w) for g, w in zip(pouts, want[:8]))
    eng.close()


@pytest.mark.parametrize("arch,C", [("fcn_skip", 3), ("fcn", 3), ("fcn_skip", 6), ("unet", 3), ("res_unet", 3)])
def test_page_units_equal_page_by_page(gpu, monkeypatch, arch, C):
    """Page slots (pseg_predict_pages_device; the units of pseg_predict_batch): every activation tensor holds one slot per page of a
    unit, the low-resolution layers take all slots in one launch (conv_sp_kernel: the tile index carries the page; conv_mfma_kernel:
    the slot is blockIdx.z), the others run per slot -- each label map must be the one pseg_predict_device gives for that page.
    Ragged shapes (pad-to-32 canvases), one-tile pages, a list that mixes shapes (units break at a shape change), uint8 and
    int64 maps, a unit size that does not divide the list.  unet / res_unet (round 5): their plain convolutions from 1/4 resolution
    down take a unit's slots in one launch (blockIdx.z), everything else -- first layer, pools, up-sampling and residual layers, the
    stand-alone logits layer -- runs per slot; the input tensor of every slot is pre-processed up front."""
    import torch
    rng = np.random.default_rng(11)
    monkeypatch.setenv("PSEG_SP_CHECK", "1")
    monkeypatch.setenv("PSEG_BATCH_PAGES", "3")
    eng = gpu.Engine(arch, C, mode=gpu.MODE_BF16)
    from pseg_amd import synth
    eng.set_weights(synth.glorot_weights(eng.weight_specs(), seed=42, gain=1.5, bias_scale=0.05))
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev).cuda_stream
    def one(p):
        d = torch.from_numpy(np.ascontiguousarray(p)).to(dev)
        lab = torch.empty(p.shape, dtype=torch.uint8, device=dev)
        eng.predict_device(d.data_ptr(), p.shape[0], p.shape[1], d_labels_u8=lab.data_ptr(), stream=st)
        torch.cuda.synchronize()
        return lab.cpu().numpy()
    for (H, W), n in (((130, 67), 7), ((300, 420), 5), ((33, 1), 4), ((512, 384), 2)):
        pages = rng.integers(0, 256, (n, H, W), dtype=np.uint8)
        want = [one(pages[i]) for i in range(n)]
        d = torch.from_numpy(pages).to(dev)
        out8 = torch.zeros((n, H, W), dtype=torch.uint8, device=dev)
        out64 = torch.zeros((n, H, W), dtype=torch.int64, device=dev)
        eng.predict_pages_device(d.data_ptr(), n, H, W, d_labels=out64.data_ptr(), d_labels_u8=out8.data_ptr(), stream=st)
        torch.cuda.synchronize()
        assert all(np.array_equal(out8[i].cpu().numpy(), want[i]) for i in range(n))
        assert all(np.array_equal(out64[i].cpu().numpy(), want[i]) for i in range(n))
    # a list of mixed shapes through the host entry: units of <= 3 same-shape neighbours
    shapes = [(96, 80)] * 4 + [(130, 67)] + [(96, 80)] * 2 + [(64, 64)] * 3
    pages = [rng.integers(0, 256, s_, dtype=np.uint8) for s_ in shapes]
    want = [one(p) for p in pages]
    got = eng.predict_batch(pages, dtype=np.uint8)
    assert all(np.array_equal(g, w) for g, w in zip(got, want))
    got = eng.predict_batch(pages, dtype=np.int64)
    assert all(np.array_equal(g, w) for g, w in zip(got, want))
    # ... and a single page afterwards still goes through the one-slot path with the same result
    assert np.array_equal(one(pages[0]), want[0])
    eng.close()


def test_config3_full_size_train_step_matches_torch_autograd(gpu):
    import oracle
    from oracle import train_ref
    from pseg_amd import synth
    H, W, C = 2048, 1536, 3
    img, _, mask = synth.synth_page(0, H, W, C)
    Wt = oracle.init_weights("fcn_skip", C, seed=3, gain=1.0, bias_scale=0.02)
    eng = gpu.Engine("fcn_skip", C, mode=gpu.MODE_F32_EXACT)
    eng.set_weights(Wt)
    eng.train_init(clipnorm=1.0)
    loss, acc, jac, dice = eng.train_forward_backward(img, mask)
    # the referee runs in float64: torch's own float32 bias-gradient reduction over 3.1 M cancelling terms is off by ~1 %
    rl, ra, rj, rd, rgrads = train_ref.fcn_loss_and_grads("fcn_skip", Wt, img, mask, float64=True)
    assert abs(loss - rl) <= 1e-4 * abs(rl), (loss, rl)               # north_star: loss within 1e-4 relative
    assert abs(acc - ra) <= 1e-5 and abs(jac - rj) <= 1e-4 * abs(rj) and abs(dice - rd) <= 1e-4 * abs(rd)
    g = eng.gradients()
    for k, want in rgrads.items():
        scale = float(np.abs(want).max()) + 1e-12
        # float32 products and partial sums over up to 3.1 M pixels against a float64 referee
        assert float(np.abs(g[k] - want).max()) <= 2e-3 * scale, k
    # >= 20 further steps (Adam lr 1e-4 is lib/network.py:23's default; 1e-3 here so that 20 steps move the loss): two engines
    # fed the same pages stay BIT-IDENTICAL -- every reduction of the step (weight / bias gradients: per-strip partial sums
    # added in strip order; loss and metrics: per-block rows added in a fixed tree; clip norms) has a fixed order since round 3
    # (rounds 1-2 accumulated with float atomics and drifted up to 5 % in the loss by step 19) -- and the loss goes down
    eng2 = gpu.Engine("fcn_skip", C, mode=gpu.MODE_F32_EXACT)
    eng2.set_weights(Wt)
    eng2.train_init(clipnorm=1.0)
    m2 = eng2.train_forward_backward(img, mask)
    assert tuple(m2) == (loss, acc, jac, dice)
    g2 = eng2.gradients()
    assert all(np.array_equal(g[k], g2[k]) for k in g), [k for k in g if not np.array_equal(g[k], g2[k])]
    pages = [(img, mask)] + [synth.synth_page(i, H, W, C)[::2] for i in (1, 2)]
    losses = [loss]
    eng.train_apply(1e-3)
    eng2.train_apply(1e-3)
    for it in range(21):
        im, mk = pages[it % len(pages)]
        l1 = eng.train_forward_backward(im, mk)[0]
        l2 = eng2.train_forward_backward(im, mk)[0]
        assert l1 == l2, (it, l1, l2)
        eng.train_apply(1e-3)
        eng2.train_apply(1e-3)
        losses.append(l1)
    assert np.isfinite(losses).all() and np.mean(losses[-3:]) < 0.8 * np.mean(losses[:3]), losses
    w1, w2 = eng.get_weights(), eng2.get_weights()
    assert all(np.array_equal(w1[k], w2[k]) for k in w1), [k for k in w1 if not np.array_equal(w1[k], w2[k])]
    eng.close()
    eng2.close()


_DP_WORKER = r'''
import os, sys, json
import numpy as np
sys.path[:0] = [%(root)r, os.path.join(%(root)r, "page-segmentation_amd")]
import torch
torch.cuda.is_available()
import torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
from pseg_amd import synth
from ocr4all_pixel_classifier.lib.network import Network
from ocr4all_pixel_classifier.lib.trainer import TrainSettings
from ocr4all_pixel_classifier.lib.dataset import Dataset, SingleData
from ocr4all_pixel_classifier.lib.colors import ColorMap
import oracle
pages = [synth.synth_page(i, 96, 128, 3) for i in range(4)]
data = [SingleData(image=p[0], binary=p[1], mask=p[2].copy(), original_shape=p[0].shape, image_path="p%%d.png" %% i) for i, p in enumerate(pages)]
cm = ColorMap({})
np.random.seed(1234)                        # every rank shuffles the page list identically (lib/network.py:134-135)
net = Network("train", n_classes=3, l_rate=1e-3)
net.model.set_weights(oracle.init_weights("fcn_skip", 3, seed=11, gain=1.0, bias_scale=0.02))
s = TrainSettings(n_epoch=3, n_classes=3, l_rate=1e-3, train_data=Dataset(data, cm), validation_data=None, display=1, threads=1,
                  output_dir=os.path.join(%(out)r, "rank%%d" %% rank), data_augmentation=False,
                  early_stopping_max_performance_drops=0)
hist = net.train_dataset(s, None, rank=rank, world=world)
w = net.model.get_weights()
np.savez(os.path.join(%(out)r, "w%%d.npz" %% rank), **{k.replace("/", "__"): v for k, v in w.items()})
json.dump(hist["loss"], open(os.path.join(%(out)r, "loss%%d.json" %% rank), "w"))
dist.destroy_process_group()
'''


def test_config3_data_parallel_two_ranks_real_network_path(gpu, tmp_path):
    """Network.train_dataset(rank, world) with two rank processes (gloo rendezvous; both use this box's one GPU): the
    replicas end bit-identical and equal a single process that averages the two pages' gradients per step."""
    import socket
    import oracle
    from pseg_amd import synth
    from pseg_amd.parallel import grad_tensor
    import torch
    script = tmp_path / "dp_worker.py"
    script.write_text(_DP_WORKER % {"root": ROOT, "out": str(tmp_path)})
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=600)[0].decode("utf8", "replace") for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    w0 = dict(np.load(tmp_path / "w0.npz"))
    w1 = dict(np.load(tmp_path / "w1.npz"))
    assert all(np.array_equal(w0[k], w1[k]) for k in w0)              # replicas bit-identical
    # single-process restatement: same shuffles, pages (2s, 2s+1) per step, gradients summed then applied with 1/2
    pages = [synth.synth_page(i, 96, 128, 3) for i in range(4)]
    order = list(range(4))
    np.random.seed(1234)
    np.random.randint(0, 2 ** 31 - 1)                                  # Network.__init__ draws the glorot seed before training
    eng = gpu.Engine("fcn_skip", 3, mode=gpu.MODE_F32_EXACT)
    eng.set_weights(oracle.init_weights("fcn_skip", 3, seed=11, gain=1.0, bias_scale=0.02))
    eng.train_init(clipnorm=1.0)
    losses = []
    for epoch in range(3):
        np.random.shuffle(order)
        ep = []
        for s in range(2):
            a, b = order[2 * s], order[2 * s + 1]
            la = eng.train_forward_backward(pages[a][0], pages[a][2])[0]
            ga = grad_tensor(eng).clone()
            lb = eng.train_forward_backward(pages[b][0], pages[b][2])[0]
            g = grad_tensor(eng)
            torch.cuda.synchronize()
            g += ga
            torch.cuda.synchronize()
            eng.train_apply(1e-3, 0.5)
            ep += [la, lb]
        losses.append(ep)
    ws = eng.get_weights()
    # every reduction of the step has a fixed order (round 3: no float atomics left) and a two-term float sum is order-free
    # (ga + gb on both sides): the replicas equal the single process that adds the two pages' gradients BIT FOR BIT
    for k, v in ws.items():
        assert np.array_equal(v, w0[k.replace("/", "__")]), (k, float(np.abs(v - w0[k.replace("/", "__")]).max()))
    eng.close()


def test_bench_self_launch_two_ranks(gpu):
    """`python bench.py --gpus 2` with WORLD_SIZE unset starts its two rank processes itself and prints one JSON line with
    n_gpus = 2 (the driver's command form).  Rehearsed on this one-GPU box with the gloo rendezvous and both ranks on cuda:0
    (PSEG_BENCH_BACKEND / PSEG_BENCH_ONE_GPU); a missing GPU makes the launcher fail fast instead of hanging."""
    import json
    env = dict(os.environ, PSEG_BENCH_BACKEND="gloo", PSEG_BENCH_ONE_GPU="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-extra",
                        "--no-cpu-baseline", "--height", "512", "--width", "384"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["steps"] == 3
    assert d["metric"] == "Mpixels/s classified (512x384, 3-class)"
    # without the rehearsal knobs rank 1 has no GPU here: non-zero exit within seconds, no hang
    env2 = dict(os.environ)
    env2.pop("WORLD_SIZE", None)
    env2.pop("RANK", None)
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-extra",
                         "--no-cpu-baseline", "--height", "64", "--width", "64"], env=env2, capture_output=True, text=True, timeout=300)
    import torch
    if torch.cuda.device_count() < 2:
        assert r2.returncode != 0 and "rank" in r2.stderr


def test_page_units_three_channel_input(gpu, monkeypatch):
    """Page units with a 3-channel page (lib/network.py:28,56 input_image_dimension = 3): no fused first layer -- every slot's input
    tensor is pre-processed up front -- and the maps equal the page-by-page ones (device entry and host list)."""
    import torch
    from pseg_amd import synth
    rng = np.random.default_rng(12)
    monkeypatch.setenv("PSEG_SP_CHECK", "1")
    monkeypatch.setenv("PSEG_BATCH_PAGES", "4")
    eng = gpu.Engine("fcn_skip", 3, in_channels=3, mode=gpu.MODE_BF16)
    eng.set_weights(synth.glorot_weights(eng.weight_specs(), seed=42, gain=1.5, bias_scale=0.05))
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev).cuda_stream
    H, W, n = 130, 200, 6
    pages = rng.integers(0, 256, (n, H, W, 3), dtype=np.uint8)
    want = [eng.predict(pages[i], want_logits=False, want_probs=False)[2] for i in range(n)]
    assert len({w.tobytes() for w in want}) == n
    d = torch.from_numpy(pages).to(dev)
    out8 = torch.zeros((n, H, W), dtype=torch.uint8, device=dev)
    eng.predict_pages_device(d.data_ptr(), n, H, W, d_labels_u8=out8.data_ptr(), stream=st)
    eng.status(st)
    assert all(np.array_equal(out8[i].cpu().numpy(), want[i]) for i in range(n))
    got = eng.predict_batch([pages[i] for i in range(n)], dtype=np.uint8)
    assert all(np.array_equal(g, w) for g, w in zip(got, want))
    eng.close()
