"""Train step (float32 engine) against the torch-CPU autograd reference + NumPy Keras-Adam
(oracle/train_ref.py).  Floating-point kernel: tolerances are written per check; north_star asks
for loss within 1e-4 relative."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _sample(seed, H, W, C):
    from pseg_amd import synth
    img, binary, mask = synth.synth_page(seed, H, W, C)
    return img, mask


@pytest.mark.gpu
@pytest.mark.parametrize("arch,C,shape", [("fcn_skip", 3, (32, 32)), ("fcn_skip", 3, (33, 17)), ("fcn", 2, (24, 200)), ("fcn_skip", 5, (200, 24))])
def test_gradients_tiny_and_ragged_pages(gpu, oracle_mod, arch, C, shape):
    """One canvas tile, pages smaller than a row piece, one-piece-wide and one-strip-tall maps: the edge cases of the strip /
    column-group walk of the weight-gradient kernels and of the first-writer-stores gradient buffers (canvas padding)."""
    from oracle.train_ref import fcn_loss_and_grads
    rng = np.random.RandomState(5)
    Wt = oracle_mod.init_weights(arch, C, seed=11, gain=1.5, bias_scale=0.05)
    img = rng.randint(0, 256, size=shape).astype(np.uint8)
    mask = rng.randint(0, C, size=shape).astype(np.uint8)
    loss_o, _, _, _, g_o = fcn_loss_and_grads(arch, Wt, img, mask)
    eng = gpu.Engine(arch, C, mode=gpu.MODE_F32_EXACT)
    eng.set_weights(Wt)
    eng.train_init(clipnorm=1.0)
    for _ in range(2):                                  # twice: the second step runs on used gradient buffers
        loss = eng.train_forward_backward(img, mask)[0]
        assert abs(loss - loss_o) <= 1e-4 * abs(loss_o)
        g = eng.gradients()
        for k in g_o:
            assert np.isfinite(g[k]).all(), k
            assert np.abs(g[k] - g_o[k]).max() <= 2e-3 * np.abs(g_o[k]).max() + 1e-9, k


@pytest.mark.gpu
@pytest.mark.parametrize("knob", ["PSEG_WGRAD_ATOMIC", "PSEG_WGRAD_NO_FLAT", "PSEG_WGRAD_NO_PAIR", "PSEG_TRAIN_ONE_STREAM", "PSEG_EXACT_REM_ANY", "PSEG_EXACT_NO_REM"])
def test_gradients_alternative_kernel_paths(gpu, oracle_mod, monkeypatch, knob):
    """The switches that select the other weight-gradient kernels (float atomics; the round-2 kernels instead of the
    flattened-row / two-source ones), the single-stream backward, and the shifted-pixel tiles of the data-gradient convs forced
    on / off: same gradients within the float bar."""
    from oracle.train_ref import fcn_loss_and_grads
    rng = np.random.RandomState(9)
    shape, C = (72, 104), 3
    Wt = oracle_mod.init_weights("fcn_skip", C, seed=11, gain=1.5, bias_scale=0.05)
    img = rng.randint(0, 256, size=shape).astype(np.uint8)
    mask = rng.randint(0, C, size=shape).astype(np.uint8)
    loss_o, _, _, _, g_o = fcn_loss_and_grads("fcn_skip", Wt, img, mask)
    monkeypatch.setenv(knob, "1")                       # an engine keeps the PSEG_* snapshot it is created under
    eng = gpu.Engine("fcn_skip", C, mode=gpu.MODE_F32_EXACT)
    monkeypatch.delenv(knob)
    eng.set_weights(Wt)
    eng.train_init(clipnorm=1.0)
    loss = eng.train_forward_backward(img, mask)[0]
    assert abs(loss - loss_o) <= 1e-4 * abs(loss_o)
    g = eng.gradients()
    for k in g_o:
        assert np.abs(g[k] - g_o[k]).max() <= 2e-3 * np.abs(g_o[k]).max() + 1e-9, (knob, k)


# (160 x 288 / 144 x 272 -- a page smaller than its canvas: several row pieces per map -- 4.5 at full, 2.25 at quarter resolution -- and row strips cut into column
# groups: the walk of the flattened-row and two-source weight-gradient kernels beyond a single piece, with ragged right edges)
@pytest.mark.parametrize("arch,C,shape", [("fcn_skip", 3, (64, 96)), ("fcn_skip", 6, (70, 50)), ("fcn", 3, (96, 64)), ("fcn_skip", 24, (64, 64)),
                                          ("fcn_skip", 3, (160, 288)), ("fcn", 4, (144, 272))])
def test_loss_metrics_and_gradients(gpu, oracle_mod, arch, C, shape):
    from oracle.train_ref import fcn_loss_and_grads
    Wt = oracle_mod.init_weights(arch, C, seed=11, gain=1.5, bias_scale=0.05)
    img, mask = _sample(2, shape[0], shape[1], C)
    loss_o, acc_o, jac_o, dice_o, g_o = fcn_loss_and_grads(arch, Wt, img, mask)
    eng = gpu.Engine(arch, C, mode=gpu.MODE_F32_EXACT)
    eng.set_weights(Wt)
    eng.train_init(clipnorm=1.0)
    loss, acc, jac, dice = eng.train_forward_backward(img, mask)
    assert abs(loss - loss_o) <= 1e-4 * abs(loss_o)              # north_star: 1e-4 relative
    assert abs(acc - acc_o) <= 2.0 / img.size                    # argmax near-ties may flip a pixel
    assert abs(jac - jac_o) <= 1e-4 and abs(dice - dice_o) <= 1e-4
    assert eng.eval_step(img, mask)[0] == pytest.approx(loss, rel=1e-6)
    g = eng.gradients()
    assert list(g.keys()) == list(g_o.keys())
    for k in g_o:
        scale = np.abs(g_o[k]).max() + 1e-12
        err = np.abs(g[k] - g_o[k]).max()
        assert err <= 2e-3 * scale + 1e-9, "%s: max err %g vs scale %g" % (k, err, scale)
    eng.close()


def test_adam_clipnorm_trajectory(gpu, oracle_mod):
    """Three steps on two pages: weights follow the NumPy restatement of Keras Adam with per-tensor
    clip_by_norm; the loss trajectory matches within 1e-4 relative."""
    from oracle.train_ref import fcn_loss_and_grads, KerasAdam
    arch, C = "fcn_skip", 3
    Wt = oracle_mod.init_weights(arch, C, seed=5, gain=1.5, bias_scale=0.05)
    pages = [_sample(s, 64, 64, C) for s in (0, 1)]
    eng = gpu.Engine(arch, C, mode=gpu.MODE_F32_EXACT)
    eng.set_weights(Wt)
    eng.train_init(clipnorm=1.0)
    opt = KerasAdam(lr=1e-3, clipnorm=1.0)
    Wo = Wt
    for step in range(3):
        img, mask = pages[step % 2]
        loss_o, _, _, _, g_o = fcn_loss_and_grads(arch, Wo, img, mask)
        loss = eng.train_forward_backward(img, mask)[0]
        assert abs(loss - loss_o) <= 1e-4 * abs(loss_o), (step, loss, loss_o)
        eng.train_apply(1e-3)
        Wo = opt.apply(Wo, g_o)
    Wg = eng.get_weights()
    for k in Wo:
        # Adam's first steps move every weight by ~lr regardless of gradient scale, so compare the
        # *update*; elements whose gradient is ~0 have an ill-conditioned m/sqrt(v) and are excluded
        d_o, d_g = Wo[k] - Wt[k], Wg[k] - Wt[k]
        assert np.abs(d_g - d_o).max() <= 0.05 * 3e-3 + 1e-7, k
        assert np.median(np.abs(d_g - d_o)) <= 2e-5, k
    # predict after training uses the updated device weights
    lab = eng.predict(pages[0][0], want_logits=False, want_probs=False)[2]
    assert lab.shape == pages[0][0].shape
    eng.close()


def test_dp_gradient_buffer_is_torch_visible(gpu, oracle_mod):
    """The flat gradient buffer can be wrapped zero-copy as a torch tensor (what the RCCL
    all-reduce operates on) and scaled averaging equals two separate steps."""
    import torch
    arch, C = "fcn", 3
    Wt = oracle_mod.init_weights(arch, C, seed=3, gain=1.5, bias_scale=0.05)
    (i0, m0), (i1, m1) = _sample(0, 64, 64, C), _sample(1, 64, 64, C)
    eng = gpu.Engine(arch, C, mode=gpu.MODE_F32_EXACT)
    eng.set_weights(Wt)
    eng.train_init()
    from pseg_amd.parallel import grad_tensor
    eng.train_forward_backward(i0, m0)
    g0 = grad_tensor(eng).clone()
    eng.train_forward_backward(i1, m1)
    gt = grad_tensor(eng)
    assert gt.is_cuda and gt.dtype == torch.float32 and gt.numel() == eng.grad_buffer()[1]
    gt += g0                                 # "all-reduce(sum)" of two ranks, in place on the engine's buffer
    torch.cuda.synchronize()
    eng.train_apply(1e-3, grad_scale=0.5)    # average
    assert np.isfinite(eng.get_weights()["logits/kernel"]).all()
    eng.close()


def test_trainer_api_end_to_end(gpu, oracle_mod, tmp_path):
    """Trainer / Network.train_dataset with the reference's settings object: loss goes down on a
    tiny synthetic set, the progress callback fires, the best checkpoint is written and reloads."""
    from pseg_amd import synth
    from ocr4all_pixel_classifier.lib.trainer import Trainer, TrainSettings
    from ocr4all_pixel_classifier.lib.dataset import Dataset, SingleData
    from ocr4all_pixel_classifier.lib.colors import ColorMap
    from ocr4all_pixel_classifier.lib.callback import TrainProgressCallback
    from ocr4all_pixel_classifier.lib.network import Network
    from ocr4all_pixel_classifier.lib.metrics import Monitor
    np.random.seed(0)
    cm = ColorMap({})

    def ds(seeds):
        out = []
        for s in seeds:
            img, binary, mask = synth.synth_page(s, 96, 96, 3)
            out.append(SingleData(image=img, binary=binary, mask=mask, original_shape=img.shape))
        return Dataset(out, cm)

    class CB(TrainProgressCallback):
        def __init__(self):
            self.total = None; self.losses = []; self.best = []
        def init(self, total_iters, early_stopping_iters):
            self.total = total_iters
        def update_loss(self, batch, loss, acc):
            self.losses.append((batch, loss, acc))
        def next_best(self, epoch, acc, n_best):
            self.best.append((epoch, acc, n_best))

    settings = TrainSettings(n_epoch=6, n_classes=3, l_rate=2e-3, train_data=ds([0, 1, 2, 3]),
                             validation_data=ds([4]), display=1, output_dir=str(tmp_path), threads=1,
                             monitor=Monitor.VAL_LOSS, evaluation_data=ds([5]))
    tr = Trainer(settings)
    cb = CB()
    hist = tr.train(cb)
    assert cb.total == 24 and len(cb.losses) == 24 and [b for b, _, _ in cb.losses] == list(range(24))
    assert len(cb.best) == 6 and len(hist["val_loss"]) == 6
    assert np.mean(hist["loss"][-2:]) < hist["loss"][0]            # it learns
    ev = tr.eval()
    assert set(ev) == {"loss", "accuracy", "jacard_coef", "dice_coef"} and np.isfinite(ev["loss"])
    assert (tmp_path / "model.h5").exists()                       # lib/network.py:177-178: <output_dir>/model.h5
    net = Network("Predict", n_classes=3, model=str(tmp_path / "model"), exact=True)
    lab = net.predict_single_data(settings.validation_data.data[0])[2]
    assert lab.shape == (96, 96)
    x, y = next(tr.train_net.create_dataset_inputs(settings.train_data, data_augmentation=False))
    assert x["input_1"].shape == (1, 96, 96, 1) and x["input_2"].shape == (1, 96, 96, 1) and y["logits"].shape == (1, 96, 96, 1)
    assert x["input_1"].dtype == np.float64 and x["input_1"].max() <= 1.0


def test_trained_weights_bf16_labels_agree_with_exact(gpu, oracle_mod):
    """random-init logits are near-tied, so argmax parity is also checked on weights trained
    for 150 steps (float32 engine, synthetic pages): the bf16 engine's label map must agree with the
    bit-exact float32 engine except at near-ties, and on the overwhelming majority of pixels."""
    from pseg_amd import synth
    e32 = gpu.Engine("fcn_skip", 3, mode=gpu.MODE_F32_EXACT)
    e32.set_weights(synth.glorot_weights(e32.weight_specs(), seed=7))
    e32.train_init(clipnorm=1.0)
    pages = [synth.synth_page(s, 128, 160, 3) for s in range(6)]
    first = last = None
    for it in range(150):
        img, _, mask = pages[it % len(pages)]
        loss = e32.train_forward_backward(img, mask)[0]
        e32.train_apply(2e-3)
        first = loss if first is None else first
        last = loss
    assert last < 0.7 * first                                    # it has learnt something
    Wt = e32.get_weights()
    eb = gpu.Engine("fcn_skip", 3, mode=gpu.MODE_BF16)
    eb.set_weights(Wt)
    img, _, mask = synth.synth_page(99, 512, 384, 3)
    z32, _, l32 = e32.predict(img, want_probs=False)
    zb, _, lb = eb.predict(img, want_probs=False)
    err = float(np.abs(zb - z32).max())
    srt = np.sort(z32, -1)
    margin = srt[..., -1] - srt[..., -2]
    assert not ((lb != l32) & (margin > 2 * err)).any()          # flips only inside the bf16 error band
    agree = float((lb == l32).mean())
    assert agree > 0.99, agree
    zo = oracle_mod.forward("fcn_skip", Wt, img[:96, :128].copy())
    z96, _, _ = e32.predict(np.ascontiguousarray(img[:96, :128]), want_probs=False)
    assert np.array_equal(zo, z96)                               # trained weights: still bit-identical to the oracle
    e32.close()
    eb.close()


@pytest.mark.parametrize("name,lr", [("sgd", 1e-2), ("rmsprop", 1e-3), ("adagrad", 1e-2), ("adadelta", 1.0),
                                     ("adamax", 2e-3), ("nadam", 2e-3), ("adam", 1e-3)])
def test_other_keras_optimizers(gpu, oracle_mod, name, lr):
    """The reference's Optimizers enum (lib/architecture.py:71-90): three steps of each update rule against
    the NumPy restatement of the TF 2.5 Keras formulas."""
    from oracle import train_ref
    from pseg_amd import synth
    Wt = oracle_mod.init_weights("fcn", 3, seed=3, gain=1.0, bias_scale=0.05)
    eng = gpu.Engine("fcn", 3, mode=gpu.MODE_F32_EXACT)
    eng.set_weights(Wt)
    eng.train_init(clipnorm=1.0)
    eng.train_set_optimizer(name)
    ref = {k: v.copy() for k, v in Wt.items()}
    opt = train_ref.KerasOptimizer(name, lr, clipnorm=1.0)
    for step in range(3):
        img, _, mask = synth.synth_page(step, 64, 64, 3)
        eng.train_forward_backward(img, mask)
        grads = eng.gradients()          # the update rule is what is under test: both sides get the same gradients
        eng.train_apply(lr)              # (the gradients themselves are checked against torch above)
        opt.apply(ref, grads)
        assert all(np.isfinite(v).all() for v in ref.values())
    got = eng.get_weights()
    for k in ref:
        scale = max(np.abs(ref[k] - Wt[k]).max(), 1e-12)
        assert np.abs(got[k] - ref[k]).max() <= 2e-4 * scale + 2e-7, (name, k)
    with pytest.raises(Exception):
        eng.train_set_optimizer("lion")
    eng.close()


@pytest.mark.parametrize("name", ["dice", "jaccard", "dice_and_crossentropy", "categorical_hinge", "categorical_focal"])
def test_alternative_losses(gpu, oracle_mod, name):
    """The reference's Loss enum (lib/metrics.py:72-133): loss value and every gradient against torch
    autograd of the same definition (bar as for cross-entropy: loss 1e-4 relative)."""
    from oracle import train_ref
    from pseg_amd import synth
    gain = 0.35 if name == "categorical_focal" else 1.0      # focal clips the raw logits to (0, 1): keep some inside
    Wt = oracle_mod.init_weights("fcn_skip", 3, seed=5, gain=gain, bias_scale=0.3 if name == "categorical_focal" else 0.05)
    img, _, mask = synth.synth_page(2, 64, 96, 3)
    eng = gpu.Engine("fcn_skip", 3, mode=gpu.MODE_F32_EXACT)
    eng.set_weights(Wt)
    eng.train_init(clipnorm=1.0)
    eng.train_set_loss(name)
    loss, acc, jac, dice = eng.train_forward_backward(img, mask)
    g = eng.gradients()
    rl, ra, rj, rd, rg = train_ref.fcn_loss_and_grads("fcn_skip", Wt, img, mask, loss_kind=name)
    assert abs(loss - rl) <= 1e-4 * max(abs(rl), 1e-3), (loss, rl)
    assert abs(acc - ra) < 1e-6 and abs(jac - rj) < 1e-5 and abs(dice - rd) < 1e-5
    nonzero = 0
    for k in rg:
        scale = np.abs(rg[k]).max()
        nonzero += scale > 0
        assert np.abs(g[k] - rg[k]).max() <= 2e-4 * scale + 1e-9, (name, k, scale)
    assert nonzero >= len(rg) - 2
    ev = eng.eval_step(img, mask)
    assert abs(ev[0] - rl) <= 1e-4 * max(abs(rl), 1e-3)
    with pytest.raises(Exception):
        eng.train_set_loss("mse")
    eng.close()


def test_c_abi_allreduce_single_rank(gpu, oracle_mod):
    """pseg_allreduce_init / pseg_train_allreduce (RCCL bound at run time) on a one-rank communicator -- what a one-GPU box
    can run: the collective initialises on the hardware, the all-reduce is enqueued on the engine's stream between the
    backward kernels and the update, and a sum over one rank leaves the gradients -- and the step that follows -- bit-identical
    to the plain step."""
    from pseg_amd import synth
    Wt = oracle_mod.init_weights("fcn_skip", 3, seed=3, gain=1.0, bias_scale=0.02)
    img, _, mask = synth.synth_page(1, 96, 128, 3)
    engines = []
    for dp in (False, True):
        e = gpu.Engine("fcn_skip", 3, mode=gpu.MODE_F32_EXACT)
        e.set_weights(Wt)
        e.train_init(clipnorm=1.0)
        if dp:
            e.allreduce_init(0, 1, gpu.Engine.allreduce_unique_id())
        engines.append(e)
    for step in range(3):
        for dp, e in zip((False, True), engines):
            e.train_forward_backward(img, mask)
            if dp:
                e.train_allreduce()
        g0, g1 = engines[0].gradients(), engines[1].gradients()
        assert all(np.array_equal(g0[k], g1[k]) for k in g0)
        for e in engines:
            e.train_apply(1e-3, 1.0)
    w0, w1 = engines[0].get_weights(), engines[1].get_weights()
    assert all(np.array_equal(w0[k], w1[k]) for k in w0)
    engines[1].allreduce_destroy()
    with pytest.raises(gpu.PsegError):
        engines[1].train_allreduce()
    for e in engines:
        e.close()


_RCCL_RANK_SCRIPT = r"""
import os, sys
import numpy as np
root = os.environ["PSEG_ROOT"]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "page-segmentation_amd"))
import pseg_amd
from pseg_amd import synth
rank, world = int(os.environ["PSEG_T_RANK"]), int(os.environ["PSEG_T_WORLD"])
uid = bytes.fromhex(os.environ["PSEG_T_UID"])
e = pseg_amd.Engine("fcn_skip", 3, device=rank, mode=pseg_amd.MODE_F32_EXACT)
e.set_weights(synth.glorot_weights(e.weight_specs(), seed=3, gain=1.0, bias_scale=0.02))
e.train_init(clipnorm=1.0)
img, _, mask = synth.synth_page(10 + rank, 96, 128, 3)        # a different page per rank (SURVEY 8e: DP over pages)
e.allreduce_init(rank, world, uid)                              # collective: every rank
e.train_forward_backward(img, mask)
own = e.gradients()
e.train_allreduce()                                             # in place, on the engine's stream
summed = e.gradients()
np.savez(os.path.join(os.environ["PSEG_T_OUT"], "rank%d.npz" % rank), **{"own/" + k: v for k, v in own.items()}, **{"sum/" + k: v for k, v in summed.items()})
e.allreduce_destroy()
e.close()
"""


def test_c_abi_allreduce_two_devices(gpu, tmp_path):
    """The C-ABI data-parallel exchange on REAL ranks: two fresh processes, one per device, pseg_allreduce_unique_id (here) ->
    pseg_allreduce_init -> pseg_train_forward_backward on a page of their own -> pseg_train_allreduce; every rank's gradient
    buffer then holds the sum of the two per-page gradients, bit for bit (a two-rank float sum is order-free) -- the averaged
    batch gradient of SURVEY 8e once pseg_train_apply scales by 1/world.  Needs two visible devices: skipped, with the reason,
    on the one-GPU box (pseg_allreduce_init with world = 1 is test_c_abi_allreduce_single_rank)."""
    import os
    import subprocess
    import sys
    ndev = gpu.device_count()
    if ndev < 2:
        pytest.skip("pseg_device_count() = %d: the two-rank RCCL path of the C ABI needs two devices" % ndev)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    uid = gpu.Engine.allreduce_unique_id()
    procs = []
    for r in range(2):
        env = dict(os.environ)
        env.update({"PSEG_ROOT": root, "PSEG_T_RANK": str(r), "PSEG_T_WORLD": "2", "PSEG_T_UID": uid.hex(), "PSEG_T_OUT": str(tmp_path),
                    "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        procs.append(subprocess.Popen([sys.executable, "-c", _RCCL_RANK_SCRIPT], env=env, stderr=subprocess.PIPE, text=True))
    errs = []
    for p in procs:
        try:
            _, err = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        errs.append(err)
    assert all(p.returncode == 0 for p in procs), [e[-1500:] for e in errs]
    ranks = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(2)]
    names = [k[4:] for k in ranks[0].files if k.startswith("own/")]
    assert names
    for k in names:
        want = ranks[0]["own/" + k] + ranks[1]["own/" + k]
        assert np.abs(want).max() > 0 or k.endswith("bias")
        for r in range(2):
            assert np.array_equal(ranks[r]["sum/" + k], want), (k, r)
