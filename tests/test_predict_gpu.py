"""Parity of the HIP predict path (through the C ABI) against the CPU oracle.

F32_EXACT mode: the GPU and the oracle run the same operation sequence (sequential fmaf chain in
(ky,kx,ci) order, + bias, ReLU), so logits must be BIT-IDENTICAL and label maps identical.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SHAPES = [(64, 96), (96, 64), (70, 50), (33, 1), (1, 37), (160, 96)]


def _page(rng, H, W):
    return rng.integers(0, 256, size=(H, W), dtype=np.uint8)


@pytest.mark.parametrize("arch,C", [("fcn_skip", 3), ("fcn_skip", 6), ("fcn", 3), ("unet", 3), ("res_unet", 3)])
def test_exact_mode_bit_identical(gpu, oracle_mod, arch, C):
    rng = np.random.default_rng(7)
    Wt = oracle_mod.init_weights(arch, C, seed=42, gain=1.5, bias_scale=0.05)
    eng = gpu.Engine(arch, C, mode=gpu.MODE_F32_EXACT)
    assert [n for n, _ in eng.weight_specs()] == list(Wt.keys())
    eng.set_weights(Wt)
    shapes = SHAPES if arch.startswith("fcn") else SHAPES[:3]
    for (H, W) in shapes:
        img = _page(rng, H, W)
        logit_o, prob_o, pred_o = oracle_mod.predict_single_data(arch, Wt, img, "f32")
        logit, prob, pred = eng.predict(img)
        assert logit.shape == (H, W, C) and pred.dtype == np.int64
        assert np.array_equal(logit, logit_o), "logits differ: max |d| = %g" % np.abs(logit - logit_o).max()
        assert np.array_equal(pred, pred_o)
        # softmax: f32 exp implementations differ by ulps -> tolerance 2e-6 absolute
        assert np.abs(prob - prob_o).max() <= 2e-6
    eng.close()


def test_exact_mode_layer_activations(gpu, oracle_mod):
    rng = np.random.default_rng(3)
    Wt = oracle_mod.init_weights("fcn_skip", 3, seed=1, gain=1.5, bias_scale=0.05)
    img = _page(rng, 70, 50)
    z, acts = oracle_mod.forward("fcn_skip", Wt, img, "f32", return_acts=True)
    eng = gpu.Engine("fcn_skip", 3, mode=gpu.MODE_F32_EXACT)
    eng.set_weights(Wt)
    eng.predict(img)
    for name, a in acts.items():
        if name == "logits":
            continue
        g = eng.activation(name)
        assert g.shape == a.shape, name
        assert np.array_equal(g, a), name
    eng.close()


def test_predict_errors(gpu, oracle_mod):
    eng = gpu.Engine("fcn_skip", 3, mode=gpu.MODE_F32_EXACT)
    with pytest.raises(gpu.PsegError):      # weights never set
        eng.predict(np.zeros((32, 32), np.uint8))
    with pytest.raises(gpu.PsegError):      # wrong shape
        eng.set_weights({"conv2d/kernel": np.zeros((3, 3, 1, 20), np.float32)})
    with pytest.raises(gpu.PsegError):      # unknown name
        eng.set_weights({"nope/kernel": np.zeros((1,), np.float32)})
    eng.close()


@pytest.mark.parametrize("mode", ["bf16", "f32"])
def test_predict_batch_matches_single_pages(gpu, oracle_mod, mode):
    """pseg_predict_batch (overlapped copies, two staging slots) == page-by-page predict, ragged sizes,
    more pages than slots, both label dtypes, empty batch."""
    rng = np.random.default_rng(3)
    shapes = [(64, 96), (33, 50), (96, 64), (64, 96), (70, 17), (128, 160), (32, 32)]
    pages = [rng.integers(0, 256, s, dtype=np.uint8) for s in shapes]
    eng = gpu.Engine("fcn_skip", 3, mode=gpu.MODE_BF16 if mode == "bf16" else gpu.MODE_F32_EXACT)
    eng.set_weights(oracle_mod.init_weights("fcn_skip", 3, seed=42, gain=1.5, bias_scale=0.05))
    want = [eng.predict(p, want_logits=False, want_probs=False)[2] for p in pages]
    got = eng.predict_batch(pages)
    got8 = eng.predict_batch(pages, dtype=np.uint8)
    assert len(got) == len(pages) and eng.predict_batch([]) == []
    for w, g, g8 in zip(want, got, got8):
        assert g.dtype == np.int64 and g8.dtype == np.uint8
        assert np.array_equal(g, w) and np.array_equal(g8, w)
    again = eng.predict_batch(pages[::-1])
    assert all(np.array_equal(a, w) for a, w in zip(again, want[::-1]))
    eng.close()
