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


@pytest.mark.parametrize("arch", ["fcn_skip", "fcn"])
def test_exact_mode_leftover_channel_tiles(gpu, oracle_mod, monkeypatch, arch):
    """Cout = 40 / 20 layers on the shifted-pixel tiles (conv_xb_kernel REM: the 8 / 4 channels behind the full cout tiles share
    a 16-row tile with the same channels of the neighbouring pixels).  Pages of a few tiles take the padded tile by default;
    PSEG_EXACT_REM_ANY sends them through the new form: logits and every activation bit-identical to the oracle, ragged pages
    included (pixel pairs / quads that straddle the right edge)."""
    rng = np.random.default_rng(11)
    Wt = oracle_mod.init_weights(arch, 3, seed=5, gain=1.5, bias_scale=0.05)
    monkeypatch.setenv("PSEG_EXACT_REM_ANY", "1")
    eng = gpu.Engine(arch, 3, mode=gpu.MODE_F32_EXACT)
    monkeypatch.delenv("PSEG_EXACT_REM_ANY")
    eng.set_weights(Wt)
    for (H, W) in [(70, 50), (64, 96), (33, 1), (1, 37), (130, 67)]:
        img = _page(rng, H, W)
        z_o, acts = oracle_mod.forward(arch, Wt, img, "f32", return_acts=True)
        z = eng.predict(img)[0]
        assert np.array_equal(z, z_o), (H, W)
        for name in ("conv2d_2", "conv2d_3", "conv2d_transpose_2"):
            assert np.array_equal(eng.activation(name), acts[name]), (H, W, name)
    eng.close()


def test_exact_mode_full_size_page_bit_identical(gpu, oracle_mod):
    """configs[1]'s page size on the float32 engine against the oracle, bit for bit: at 2048x1536 every layer takes its
    full-size plan (8-row tiles, unsplit cout blocks, the left-over channel tiles of conv3 / conv4 / deconv3 by default)."""
    rng = np.random.default_rng(21)
    Wt = oracle_mod.init_weights("fcn_skip", 3, seed=9, gain=1.5, bias_scale=0.05)
    img = _page(rng, 2048, 1536)
    z_o, acts = oracle_mod.forward("fcn_skip", Wt, img, "f32", return_acts=True)
    eng = gpu.Engine("fcn_skip", 3, mode=gpu.MODE_F32_EXACT)
    eng.set_weights(Wt)
    z, _, pred = eng.predict(img, want_probs=False)
    assert np.array_equal(z, z_o)
    assert np.array_equal(pred, np.argmax(z_o, -1))
    for name in ("conv2d_1", "conv2d_2", "conv2d_3", "conv2d_6", "conv2d_transpose_2"):
        assert np.array_equal(eng.activation(name), acts[name]), name
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


@pytest.mark.parametrize("mode", ["bf16", "f32"])
def test_call_sequences_do_not_leak_state(gpu, oracle_mod, mode):
    """One engine driven through a random sequence of host / device / batch predicts, canvas growth and
    shrinkage and weight changes must return exactly what a fresh engine returns for each input."""
    import torch
    m = gpu.MODE_BF16 if mode == "bf16" else gpu.MODE_F32_EXACT
    rng = np.random.default_rng(17)
    shapes = [(64, 96), (1024, 1536), (33, 50), (512, 384), (2048, 1024), (96, 64)]
    pages = {s: rng.integers(0, 256, s, dtype=np.uint8) for s in shapes}
    weights = [oracle_mod.init_weights("fcn_skip", 3, seed=sd, gain=1.5, bias_scale=0.05) for sd in (1, 2)]
    want = {}
    for wi, Wt in enumerate(weights):
        for s in shapes:
            f = gpu.Engine("fcn_skip", 3, mode=m)
            f.set_weights(Wt)
            want[(wi, s)] = f.predict(pages[s], want_probs=False)
            f.close()
    eng = gpu.Engine("fcn_skip", 3, mode=m)
    wi = 0
    eng.set_weights(weights[wi])
    st = torch.cuda.current_stream().cuda_stream
    for step in range(40):
        op = int(rng.integers(0, 5))
        s = shapes[int(rng.integers(0, len(shapes)))]
        if op == 0:                                              # change the weights
            wi = 1 - wi
            eng.set_weights(weights[wi])
        elif op == 1:                                            # host entry with logits
            z, _, l = eng.predict(pages[s], want_probs=False)
            assert np.array_equal(z, want[(wi, s)][0]) and np.array_equal(l, want[(wi, s)][2]), (step, s)
        elif op == 2:                                            # host entry, labels only
            l = eng.predict(pages[s], want_logits=False, want_probs=False)[2]
            assert np.array_equal(l, want[(wi, s)][2]), (step, s)
        elif op == 3:                                            # device entry on torch's stream, uint8 labels
            d_img = torch.from_numpy(pages[s]).cuda()
            d_u8 = torch.empty(s, dtype=torch.uint8, device="cuda")
            eng.predict_device(d_img.data_ptr(), s[0], s[1], d_labels_u8=d_u8.data_ptr(), stream=st)
            torch.cuda.synchronize()
            assert np.array_equal(d_u8.cpu().numpy(), want[(wi, s)][2]), (step, s)
        else:                                                    # batch of three ragged pages
            ss = [shapes[int(i)] for i in rng.integers(0, len(shapes), 3)]
            out = eng.predict_batch([pages[q] for q in ss])
            for q, o in zip(ss, out):
                assert np.array_equal(o, want[(wi, q)][2]), (step, q)
    eng.close()


@pytest.mark.parametrize("arch", ["fcn_skip", "unet", "res_unet"])
def test_three_channel_input(gpu, oracle_mod, arch):
    """input_image_dimension = 3 (lib/network.py:28,56; RGB pages): the exact engine stays bit-identical, the bf16
    engine within its bars, and the train step matches the restated graph's loss."""
    rng = np.random.default_rng(17)
    C = 3
    Wt = oracle_mod.init_weights(arch, C, seed=6, in_ch=3, gain=1.5, bias_scale=0.05)
    img = rng.integers(0, 256, size=(70, 50, 3), dtype=np.uint8)
    z_o, _, l_o = oracle_mod.predict_single_data(arch, Wt, img, "f32")
    eng = gpu.Engine(arch, C, in_channels=3, mode=gpu.MODE_F32_EXACT)
    eng.set_weights(Wt)
    z, _, lab = eng.predict(img)
    assert np.array_equal(z, z_o) and np.array_equal(lab, l_o)
    eng.close()
    zb_o = oracle_mod.forward(arch, Wt, img, "bf16")
    eb = gpu.Engine(arch, C, in_channels=3, mode=gpu.MODE_BF16)
    eb.set_weights(Wt)
    zb, _, lb = eb.predict(img)
    assert np.abs(zb - zb_o).max() <= 2e-2 * max(1.0, np.abs(zb_o).max())
    assert np.array_equal(lb, np.argmax(zb, -1))
    with pytest.raises(gpu.PsegError):
        eb.predict(img[..., 0])                                   # a gray page for a 3-channel engine
    eb.close()
