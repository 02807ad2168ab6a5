"""bf16 MFMA throughput mode vs the bf16-emulating oracle (same rounding points: bf16 kernels,
bf16 layer outputs, float32 accumulation).  MFMA sums each 32-wide k-step in hardware order, so
float32 accumulation differs from the oracle's sequential chain by rounding noise; an output
that lands within that noise of a bf16 rounding boundary flips by one bf16 ulp (2^-8 relative).
Tolerances (stated per check): activations / logits within 2 % of the tensor's max magnitude;
label maps identical except at pixels whose oracle top-2 logit margin is within twice the observed
logit error (the only place an argmax can legitimately flip)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 0.02


def _check_labels(pred, logit, logit_o):
    pred_o = np.argmax(logit_o, -1)
    srt = np.sort(logit_o, -1)
    margin = srt[..., -1] - srt[..., -2]
    err = float(np.abs(logit - logit_o).max())
    bad = (pred != pred_o) & (margin > 2 * err + 1e-6)
    return int(bad.sum()), int((pred != pred_o).sum())


@pytest.mark.parametrize("arch,C,shape", [
    ("fcn_skip", 3, (64, 96)), ("fcn_skip", 3, (70, 50)), ("fcn_skip", 6, (160, 96)), ("fcn_skip", 3, (33, 1)),
    ("fcn_skip", 3, (256, 320)), ("fcn", 3, (96, 64)), ("unet", 3, (64, 96)), ("res_unet", 3, (70, 50)),
    ("fcn_skip", 20, (64, 96)), ("unet", 40, (32, 64)),          # more classes than one MFMA tile has rows
])
def test_bf16_mode_vs_bf16_oracle(gpu, oracle_mod, monkeypatch, arch, C, shape):
    rng = np.random.default_rng(11)
    H, W = shape
    img = rng.integers(0, 256, size=(H, W), dtype=np.uint8)
    Wt = oracle_mod.init_weights(arch, C, seed=42, gain=1.5, bias_scale=0.05)
    z_o, acts = oracle_mod.forward(arch, Wt, img, "bf16", return_acts=True)
    # the default engine: final outputs
    eng = gpu.Engine(arch, C, mode=gpu.MODE_BF16)
    eng.set_weights(Wt)
    logit, prob, pred = eng.predict(img)
    eng.close()
    # an engine that keeps the tensors the default one never writes (pool-only conv outputs, the skip into the
    # logits layer): intermediate activations layer by layer
    monkeypatch.setenv("PSEG_NO_POOL_ONLY", "1")
    monkeypatch.setenv("PSEG_NO_SKIPLOG", "1")
    monkeypatch.setenv("PSEG_NO_TAIL2", "1")
    monkeypatch.setenv("PSEG_NO_RELU_FWD", "1")      # res_unet: tensors otherwise stored after their readers' pre-activation ReLU
    monkeypatch.setenv("PSEG_NO_DQ", "1")            # fcn graphs: deconv1's output otherwise lives only inside the kernel that also runs deconv2
    eng = gpu.Engine(arch, C, mode=gpu.MODE_BF16)
    eng.set_weights(Wt)
    logit_k, _, pred_k = eng.predict(img)
    # same products; another float32 summation order may flip a bf16 rounding of the in-tail deconv here and there
    assert np.abs(logit_k - logit).max() <= 2e-3 * max(1.0, np.abs(logit).max())
    checked = 0
    for name, a in acts.items():
        if name == "logits":
            continue
        try:
            g = eng.activation(name)
        except gpu.PsegError as ex:      # tensors that live only inside a fused kernel
            assert "fused" in str(ex), ex
            continue
        checked += 1
        assert g.shape == a.shape, name
        err = np.abs(g - a).max()
        assert err <= TOL * max(1.0, np.abs(a).max()), "%s: max err %g (max |a| %g)" % (name, err, np.abs(a).max())
    assert checked >= len(acts) - 3      # logits + at most two tensors that live only inside fused kernels
    # whole-net logits: the 19- and 23-conv graphs collect more one-ulp bf16 flips (MFMA sums a k-step in hardware order and
    # starts from the bias; the oracle adds the bias last) than the 13-layer fcn graphs: 3 % there, as tests/test_api_gpu.py
    logit_tol = 0.03 if arch in ("unet", "res_unet") else TOL
    assert np.abs(logit - z_o).max() <= logit_tol * max(1.0, np.abs(z_o).max())
    bad, total = _check_labels(pred, logit, z_o)
    assert np.array_equal(pred, np.argmax(logit, -1))
    assert bad == 0, "%d label mismatches outside near-ties (%d total)" % (bad, total)
    assert np.abs(prob.sum(-1) - 1).max() < 1e-5
    eng.close()


def test_bf16_default_plan_full_size_page_vs_bf16_oracle(gpu, oracle_mod):
    """configs[1]'s page -- 2048x1536, 3 classes, fcn_skip, bf16 -- on the DEFAULT engine against the bf16-emulating oracle, directly: at
    this size the engine takes the plans the headline number is measured on (conv12_ws_kernel's persistent walk of 24 tiles per
    CU, conv_pp_kernel for conv3 / conv4, conv_sp_kernel for conv7 and deconv1 + deconv2, three-workgroup instances for the
    quarter-resolution layers, tail_fused2_kernel), which the small-page tests reach only through bit-identity chains.  Same bars
    as test_bf16_mode_vs_bf16_oracle: logits and every stored activation within 2 % of the tensor's magnitude, labels equal to the
    oracle's except where its top-2 margin is within twice the observed logit error, labels = argmax of the returned logits."""
    from pseg_amd import synth
    img = synth.synth_page(1000, 2048, 1536, 3)[0]
    Wt = oracle_mod.init_weights("fcn_skip", 3, seed=42, gain=1.5, bias_scale=0.05)
    z_o, acts = oracle_mod.forward("fcn_skip", Wt, img, "bf16", return_acts=True)
    eng = gpu.Engine("fcn_skip", 3, mode=gpu.MODE_BF16)
    eng.set_weights(Wt)
    logit, _, pred = eng.predict(img, want_probs=False)
    assert np.abs(logit - z_o).max() <= TOL * max(1.0, np.abs(z_o).max())
    bad, total = _check_labels(pred, logit, z_o)
    assert np.array_equal(pred, np.argmax(logit, -1))
    assert bad == 0, "%d label mismatches outside near-ties (%d total)" % (bad, total)
    assert total < 0.02 * pred.size                                   # near-ties are rare even with random weights
    checked = 0
    for name, a in acts.items():
        if name == "logits":
            continue
        try:
            g = eng.activation(name)
        except gpu.PsegError as ex:                                   # tensors the default plan never writes
            assert "fused" in str(ex), ex
            continue
        checked += 1
        err = np.abs(g - a).max()
        assert err <= TOL * max(1.0, np.abs(a).max()), "%s: max err %g (max |a| %g)" % (name, err, np.abs(a).max())
    assert checked >= 5                                               # conv3, conv5, conv6 and their pools, deconv2, deconv3
    eng.close()


def test_bf16_first_layer(gpu, oracle_mod, monkeypatch):
    """conv1 (MFMA, K = 25 taps): float32 sums of exact bf16 products; only the summation order
    differs from the oracle's sequential chain, so after bf16 rounding almost every value is
    identical and none is off by more than one bf16 ulp (2^-7 relative)."""
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, size=(70, 50), dtype=np.uint8)
    Wt = oracle_mod.init_weights("fcn_skip", 3, seed=3, gain=1.5, bias_scale=0.05)
    _, acts = oracle_mod.forward("fcn_skip", Wt, img, "bf16", return_acts=True)
    monkeypatch.setenv("PSEG_NO_CONV1_FUSION", "1")     # materialise conv1 (stand-alone MFMA kernel)
    eng = gpu.Engine("fcn_skip", 3, mode=gpu.MODE_BF16)
    eng.set_weights(Wt)
    eng.predict(img, want_logits=False, want_probs=False)
    g, a = eng.activation("conv2d"), acts["conv2d"]
    assert (g == a).mean() > 0.99
    assert np.all(np.abs(g - a) <= np.abs(a) * 2.0 ** -7 + 1e-30)
    eng.close()


def test_bf16_device_entry_and_canvas_reuse(gpu, oracle_mod):
    """pseg_predict_device with torch-owned buffers; shrinking then growing pages reuse buffers."""
    import torch
    rng = np.random.default_rng(9)
    Wt = oracle_mod.init_weights("fcn_skip", 3, seed=42, gain=1.5, bias_scale=0.05)
    eng = gpu.Engine("fcn_skip", 3, mode=gpu.MODE_BF16)
    eng.set_weights(Wt)
    for (H, W) in [(128, 96), (64, 64), (128, 96), (96, 160)]:
        img = rng.integers(0, 256, size=(H, W), dtype=np.uint8)
        _, _, pred_host = eng.predict(img, want_logits=False, want_probs=False)
        t_img = torch.from_numpy(img).cuda()
        t_lab = torch.empty((H, W), dtype=torch.uint8, device="cuda")
        t_l64 = torch.empty((H, W), dtype=torch.int64, device="cuda")
        eng.predict_device(t_img.data_ptr(), H, W, d_labels=t_l64.data_ptr(), d_labels_u8=t_lab.data_ptr(),
                           stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(t_lab.cpu().numpy(), pred_host.astype(np.uint8))
        assert np.array_equal(t_l64.cpu().numpy(), pred_host)
    eng.close()


@pytest.mark.parametrize("arch,C", [("fcn_skip", 6), ("fcn_skip", 3), ("fcn_skip", 11), ("fcn", 3), ("fcn", 7)])
def test_bf16_tail_variants_agree(gpu, oracle_mod, monkeypatch, arch, C):
    """Three ways to run deconv5 -> logits -> argmax: (c) composed (default: deconv5 o logits folded into
    one weight matrix on the host, bf16-rounded), (f) fused (deconv5 tile kept in registers, bf16-rounded,
    then the logits MFMA), (u) separate deconv / logits kernels.  (f) and (u) share their rounding points
    and agree to float32 noise; (c) rounds the folded kernel instead of the deconv5 activations: logits
    within 0.5 % of the largest logit, labels identical except inside that error band."""
    rng = np.random.default_rng(21)
    img = rng.integers(0, 256, size=(96, 130), dtype=np.uint8)
    Wt = oracle_mod.init_weights(arch, C, seed=4, gain=1.5, bias_scale=0.05)

    def run():
        eng = gpu.Engine(arch, C, mode=gpu.MODE_BF16)
        eng.set_weights(Wt)
        out = eng.predict(img)
        return eng, out

    eng, (z_c, p_c, l_c) = run()
    eng.close()
    monkeypatch.setenv("PSEG_NO_TAIL_COMPOSE", "1")
    eng, (z_f, p_f, l_f) = run()
    eng.close()
    monkeypatch.setenv("PSEG_NO_TAIL_FUSION", "1")
    eng, (z_u, p_u, l_u) = run()
    d5 = eng.activation("conv2d_transpose_4")
    eng.close()
    assert d5.shape[2] == 20 and np.abs(d5).max() > 0
    assert np.abs(z_f - z_u).max() <= 1e-4 * max(1.0, np.abs(z_u).max())
    assert np.abs(p_f - p_u).max() <= 1e-4
    assert _check_labels(l_f, z_f, z_u)[0] == 0
    assert np.abs(z_c - z_u).max() <= 5e-3 * max(1.0, np.abs(z_u).max())
    assert np.abs(p_c - p_u).max() <= 2e-2 and np.abs(p_c.sum(-1) - 1).max() < 1e-5
    assert np.array_equal(l_c, np.argmax(z_c, -1))
    assert _check_labels(l_c, z_c, z_u)[0] == 0


def test_bf16_conv1_fusion_is_bit_identical_to_unfused(gpu, oracle_mod, monkeypatch):
    """conv1 recomputed inside conv2's workgroup uses the same k order and rounding points as the
    stand-alone first-layer kernel: logits of the fused and unfused engines are identical."""
    rng = np.random.default_rng(33)
    Wt = oracle_mod.init_weights("fcn_skip", 3, seed=8, gain=1.5, bias_scale=0.05)
    monkeypatch.setenv("PSEG_NO_SKIPLOG", "1")        # keep conv2's tensor in memory: it is compared below
    monkeypatch.setenv("PSEG_NO_PAIRC2", "1")         # the k-chunk order of the unfused kernel (the default fused order: next test)
    monkeypatch.setenv("PSEG_NO_C32", "1")            # ... and its 16x16x32 first layer (the default 32x32x16 form sums in another order)
    outs = []
    for fuse in (True, False):
        if not fuse:
            monkeypatch.setenv("PSEG_NO_CONV1_FUSION", "1")
        eng = gpu.Engine("fcn_skip", 3, mode=gpu.MODE_BF16)
        eng.set_weights(Wt)
        res = []
        for (H, W) in [(70, 50), (160, 224), (33, 1)]:
            img = np.random.default_rng(H).integers(0, 256, size=(H, W), dtype=np.uint8)
            z, _, lab = eng.predict(img, want_probs=False)
            res.append((z, lab, eng.activation("conv2d_1")))
        if fuse:
            with pytest.raises(gpu.PsegError):
                eng.activation("conv2d")            # never materialised
        eng.close()
        outs.append(res)
    for (zf, lf, cf), (zu, lu, cu) in zip(*outs):
        assert np.array_equal(cf, cu) and np.array_equal(zf, zu) and np.array_equal(lf, lu)


@pytest.mark.parametrize("C,shape", [(3, (70, 50)), (6, (160, 224)), (3, (33, 1)), (8, (96, 130))])
def test_bf16_skip_logits_fusion_agrees_with_the_stored_skip_tensor(gpu, oracle_mod, monkeypatch, C, shape):
    """fcn_skip: conv2 hands the logits layer its contribution (4 / 8 floats per pixel, taken from the same bf16-rounded
    values) instead of storing its 30-channel full-resolution tensor.  Same products, another float32 summation
    order: logits agree to float32 noise, labels except at exact near-ties; the tensor is reported as fused."""
    Wt = oracle_mod.init_weights("fcn_skip", C, seed=9, gain=1.5, bias_scale=0.05)
    img = np.random.default_rng(shape[0]).integers(0, 256, size=shape, dtype=np.uint8)
    eng = gpu.Engine("fcn_skip", C, mode=gpu.MODE_BF16)
    eng.set_weights(Wt)
    z1, p1, l1 = eng.predict(img)
    with pytest.raises(gpu.PsegError, match="fused"):
        eng.activation("conv2d_1")
    assert eng.activation("max_pooling2d").shape[2] == 30          # the pooled tensor is still written
    eng.close()
    monkeypatch.setenv("PSEG_NO_SKIPLOG", "1")
    eng = gpu.Engine("fcn_skip", C, mode=gpu.MODE_BF16)
    eng.set_weights(Wt)
    z0, p0, l0 = eng.predict(img)
    assert eng.activation("conv2d_1").shape[2] == 30
    eng.close()
    assert np.abs(z1 - z0).max() <= 2e-6 * max(1.0, np.abs(z0).max())
    assert np.abs(p1 - p0).max() <= 1e-5
    assert _check_labels(l1, z1, z0)[0] == 0 and np.array_equal(l1, np.argmax(z1, -1))


@pytest.mark.parametrize("arch", ["fcn_skip", "fcn"])
@pytest.mark.parametrize("C,shape", [(3, (70, 50)), (6, (160, 224)), (3, (96, 130))])
def test_bf16_inner_deconv_inside_the_tail(gpu, oracle_mod, monkeypatch, arch, C, shape):
    """fcn_skip / fcn: deconv4 (k2 s2, ReLU), read by nothing but the composed tail, is recomputed inside the tail kernel (one
    sub-pixel parity per wave) instead of being stored: the bf16-rounded activations are the same values, so the logits
    agree with the separate-kernel path to float32 noise; the tensor is reported as fused."""
    Wt = oracle_mod.init_weights(arch, C, seed=13, gain=1.5, bias_scale=0.05)
    img = np.random.default_rng(shape[1]).integers(0, 256, size=shape, dtype=np.uint8)
    eng = gpu.Engine(arch, C, mode=gpu.MODE_BF16)
    eng.set_weights(Wt)
    z1, p1, l1 = eng.predict(img)
    with pytest.raises(gpu.PsegError, match="fused"):
        eng.activation("conv2d_transpose_3")
    eng.close()
    monkeypatch.setenv("PSEG_NO_TAIL2", "1")
    eng = gpu.Engine(arch, C, mode=gpu.MODE_BF16)
    eng.set_weights(Wt)
    z0, p0, l0 = eng.predict(img)
    assert eng.activation("conv2d_transpose_3").shape[2] == 30
    eng.close()
    # a deconv4 value that sits on a bf16 rounding boundary may round the other way (another float32 summation order):
    # rare single-ulp differences of one input channel, far inside the bf16 bar
    assert np.abs(z1 - z0).max() <= 2e-3 * max(1.0, np.abs(z0).max())
    assert np.mean(np.abs(z1 - z0) > 1e-5 * max(1.0, np.abs(z0).max())) < 0.02
    assert _check_labels(l1, z1, z0)[0] == 0 and np.array_equal(l1, np.argmax(z1, -1))


def test_bf16_fused_paths_random_page_sizes(gpu, oracle_mod, monkeypatch):
    """The default fcn_skip engine (skip logits from conv2, deconv4 inside the tail, pool-only stores dropped, XCD tile
    order) against an engine with all of that switched off, on page sizes that exercise every tile edge."""
    rng = np.random.default_rng(2024)
    shapes = [(1, 1), (1, 65), (31, 33), (32, 32), (33, 31), (47, 129), (64, 64), (65, 97), (100, 7), (129, 200), (200, 45)]
    for C in (3, 6):
        Wt = oracle_mod.init_weights("fcn_skip", C, seed=C, gain=1.5, bias_scale=0.05)
        imgs = [rng.integers(0, 256, size=s, dtype=np.uint8) for s in shapes]
        fused = gpu.Engine("fcn_skip", C, mode=gpu.MODE_BF16)
        fused.set_weights(Wt)
        outs = [fused.predict(im, want_probs=False) for im in imgs]
        fused.close()
        for k in ("PSEG_NO_TAIL2", "PSEG_NO_SKIPLOG", "PSEG_NO_POOL_ONLY", "PSEG_NO_XCD"):
            monkeypatch.setenv(k, "1")
        plain = gpu.Engine("fcn_skip", C, mode=gpu.MODE_BF16)
        plain.set_weights(Wt)
        for im, (z1, _, l1) in zip(imgs, outs):
            z0, _, l0 = plain.predict(im, want_probs=False)
            assert z1.shape == z0.shape == im.shape + (C,)
            assert np.abs(z1 - z0).max() <= 2e-3 * max(1.0, np.abs(z0).max()), im.shape
            assert _check_labels(l1, z1, z0)[0] == 0 and np.array_equal(l1, np.argmax(z1, -1)), im.shape
        plain.close()
        for k in ("PSEG_NO_TAIL2", "PSEG_NO_SKIPLOG", "PSEG_NO_POOL_ONLY", "PSEG_NO_XCD"):
            monkeypatch.delenv(k)


@pytest.mark.parametrize("arch", ["fcn_skip", "fcn"])
def test_bf16_wave_specialised_conv12_is_bit_identical_to_the_fused_instance(gpu, oracle_mod, monkeypatch, arch):
    """conv12_ws_kernel (producer waves recompute conv1 into a double-buffered LDS tile, consumer waves run conv2's
    k-loop) against the every-wave-does-everything fused instance (PSEG_NO_WS): with the same k order and packing
    (PSEG_NO_PAIRC2) the same bits; with its default packing (channels 16-19 of two neighbouring pixels in one k-chunk)
    the same products in another order -- on page sizes around every tile edge and on a page with several tiles per
    workgroup."""
    rng = np.random.default_rng(77)
    shapes = [(1, 1), (16, 32), (17, 33), (31, 65), (96, 80), (100, 7), (129, 200), (512, 384), (1100, 900)]
    for C in (3, 6):
        Wt = oracle_mod.init_weights(arch, C, seed=C + 1, gain=1.5, bias_scale=0.05)
        imgs = [rng.integers(0, 256, size=s, dtype=np.uint8) for s in shapes]
        def run_all():
            eng = gpu.Engine(arch, C, mode=gpu.MODE_BF16)
            eng.set_weights(Wt)
            res = []
            for im in imgs:
                z, _, l = eng.predict(im, want_probs=False)
                res.append((z, l, eng.activation("max_pooling2d")))
            eng.close()
            return res
        paired = run_all()                                    # default: 17 k-steps (paired half chunks), 32x32x16 first layer
        for form in ("1", "2"):                               # the consumers' other tile loops: same sums, same bits
            monkeypatch.setenv("PSEG_WS_FORM", form)
            other = run_all()
            monkeypatch.delenv("PSEG_WS_FORM")
            for (z1, l1, p1), (z2, l2, p2) in zip(paired, other):
                assert np.array_equal(p1, p2) and np.array_equal(z1, z2) and np.array_equal(l1, l2), form
        monkeypatch.setenv("PSEG_NO_PAIRC2", "1")
        monkeypatch.setenv("PSEG_NO_C32", "1")
        plain = run_all()                                     # the fused instance's 19-step order and 16x16x32 first layer
        monkeypatch.setenv("PSEG_NO_WS", "1")
        ref = run_all()
        monkeypatch.delenv("PSEG_NO_WS")
        monkeypatch.delenv("PSEG_NO_PAIRC2")
        monkeypatch.delenv("PSEG_NO_C32")
        for im, (z1, l1, p1), (z0, l0, p0), (zp, lp, pp) in zip(imgs, plain, ref, paired):
            assert np.array_equal(p1, p0) and np.array_equal(z1, z0) and np.array_equal(l1, l0), im.shape
            # paired half chunks: the same products in another float32 summation order -- a conv2 output may round to the
            # neighbouring bf16 value here and there, nothing more
            assert np.all(np.abs(pp - p0) <= np.abs(p0) * 2.0 ** -7 + 1e-6), im.shape
            assert (pp != p0).mean() < 0.02, im.shape
            assert np.abs(zp - z0).max() <= 1e-2 * max(1.0, float(np.abs(z0).max())), im.shape


@pytest.mark.parametrize("shape", [(64, 96), (70, 50), (160, 224)])
def test_bf16_res_unet_plan_variants_are_bit_identical(gpu, oracle_mod, monkeypatch, shape):
    """res_unet's bf16 plan moves work without changing a value: (1) a conv output read only through pre-activation ReLUs
    is stored ReLU'd by its producer (PSEG_NO_RELU_FWD keeps the raw tensor and the readers' in-LDS ReLU pass); (2) the
    stride-2 encoder convs stage their halo tile with the columns de-interleaved by parity and run four-row tiles
    (PSEG_NO_S2_MT2: eight-row tiles).  Same products in the same k order: the logits are the same bits."""
    from pseg_amd import synth
    C = 3
    Wt = oracle_mod.init_weights("res_unet", C, seed=9, gain=1.5, bias_scale=0.05)
    img = synth.synth_page(8, max(shape[0], 96), max(shape[1], 96), C)[0][:shape[0], :shape[1]].copy()

    def run():
        e = gpu.Engine("res_unet", C, mode=gpu.MODE_BF16)
        e.set_weights(Wt)
        z, _, l = e.predict(img, want_probs=False)
        return e, z, l
    e0, z0, l0 = run()
    with pytest.raises(gpu.PsegError, match="fused"):
        e0.activation("conv2d")          # the stem's first conv: stored after conv_block's ReLU
    assert e0.activation("conv2d_2").shape[2] == 32          # the shortcut is not
    e0.close()
    for knob in ("PSEG_NO_RELU_FWD", "PSEG_NO_RELU_COPY", "PSEG_NO_S2_MT2"):
        monkeypatch.setenv(knob, "1")
        e1, z1, l1 = run()
        monkeypatch.delenv(knob)
        assert np.array_equal(z1, z0) and np.array_equal(l1, l0), knob
        if knob == "PSEG_NO_RELU_FWD":
            assert e1.activation("conv2d").min() < 0
        e1.close()
    z_o = oracle_mod.forward("res_unet", Wt, img, "f32")
    assert np.abs(z0 - z_o).max() <= 0.03 * max(1.0, np.abs(z_o).max())


@pytest.mark.parametrize("arch,shape", [("fcn_skip", (1024, 768)), ("fcn_skip", (1100, 1300)), ("fcn", (1056, 1000))])
def test_bf16_ping_pong_mid_layer_kernel_is_bit_identical(gpu, monkeypatch, arch, shape):
    """conv_pp_kernel (conv3 / conv4 on pages with >= 2 tiles per CU: resident weights, producer waves, two consumer teams
    alternating k-loop and epilogue) against the three-workgroup instances of conv_mfma_kernel (PSEG_NO_PP): same packing,
    same k order, same start value -- the same bits, also in the tensors the two layers write."""
    from pseg_amd import synth
    img = synth.synth_page(5, shape[0], shape[1], 3)[0]
    res = []
    for knob in (None, "PSEG_NO_PP"):
        if knob:
            monkeypatch.setenv(knob, "1")
        e = gpu.Engine(arch, 3, mode=gpu.MODE_BF16)
        e.set_weights(synth.glorot_weights(e.weight_specs(), seed=42, gain=1.5, bias_scale=0.05))
        z, _, l = e.predict(img, want_probs=False)
        res.append((z, l, e.activation("conv2d_2"), e.activation("max_pooling2d_1")))
        e.close()
        if knob:
            monkeypatch.delenv(knob)
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a, b)
    assert np.abs(res[0][2]).max() > 0


@pytest.mark.parametrize("arch,C,shape", [("fcn_skip", 3, (96, 80)), ("fcn_skip", 6, (130, 67)), ("fcn", 3, (300, 420)), ("fcn_skip", 3, (1056, 1000)),
                                          ("fcn_skip", 3, (33, 1)), ("fcn", 6, (1, 37))])
def test_bf16_streamed_weights_kernel_is_bit_identical(gpu, monkeypatch, arch, C, shape):
    """conv_sp_kernel (conv5, conv6, conv7, deconv1 + deconv2, deconv3: persistent workgroups, loader waves streaming the weights
    through an LDS ring and staging the channel blocks of the halo tile, compute waves that never wait at a barrier) against the
    conv_mfma_kernel instances it replaces (PSEG_NO_SP): same packing, same k order, same start value -- the same bits in the
    logits, the labels and every tensor these layers write, on one-tile pages, ragged edges, several tiles per workgroup and six
    classes.  PSEG_SP_ALL routes every eligible layer to it (the release picks per layer and page size); PSEG_SP_CHECK turns a
    counter wait that gave up -- the kernel's polling loops are bounded -- into an error instead of a wrong result."""
    from pseg_amd import synth
    img = synth.synth_page(9, shape[0], shape[1], C)[0] if min(shape) >= 64 else np.random.default_rng(9).integers(0, 256, shape, dtype=np.uint8)
    res = []
    for env in ({"PSEG_SP_ALL": "1", "PSEG_SP_CHECK": "1"}, {"PSEG_SP_CHECK": "1"}, {"PSEG_NO_SP": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        e = gpu.Engine(arch, C, mode=gpu.MODE_BF16)
        e.set_weights(synth.glorot_weights(e.weight_specs(), seed=42, gain=1.5, bias_scale=0.05))
        z, _, l = e.predict(img, want_probs=False)
        z2, _, l2 = e.predict(img, want_probs=False)      # (a second page through the same engine: plans, counters and rings start over)
        assert np.array_equal(z, z2) and np.array_equal(l, l2)
        acts = [z, l]
        for name in ("conv2d_4", "conv2d_5", "conv2d_6", "conv2d_transpose_1", "conv2d_transpose_2"):
            try:
                acts.append(e.activation(name))
            except Exception:
                acts.append(None)                         # (fused away in this graph: the same in every run)
        res.append(acts)
        e.close()
        for k in env:
            monkeypatch.delenv(k)
    for other in res[1:]:
        for a, b in zip(res[0], other):
            assert (a is None) == (b is None)
            if a is not None:
                assert np.array_equal(a, b)
    assert np.abs(res[0][0]).max() > 0


@pytest.mark.parametrize("arch,shape", [("fcn_skip", (300, 420)), ("unet", (96, 160)), ("res_unet", (70, 50)), ("fcn", (1056, 1000))])
def test_bf16_epilogue_store_patch_is_bit_identical(gpu, monkeypatch, arch, shape):
    """The conv epilogue's stores through LDS (packed tiles written into a patch laid out like the tensor, read back as
    whole-line 16-byte pieces) against the direct 8-byte stores (PSEG_NO_LDS_STORE): the same bytes -- logits, labels and
    every stored activation, ragged page edges and channel counts that do not fill their last cout tile included."""
    from pseg_amd import synth
    img = synth.synth_page(7, shape[0], shape[1], 3)[0]
    res = []
    for knob in (None, "PSEG_NO_LDS_STORE"):
        if knob:
            monkeypatch.setenv(knob, "1")
        e = gpu.Engine(arch, 3, mode=gpu.MODE_BF16)
        e.set_weights(synth.glorot_weights(e.weight_specs(), seed=42, gain=1.5, bias_scale=0.05))
        z, _, l = e.predict(img, want_probs=False)
        acts = []
        for name in dict.fromkeys(spec[0].split('/')[0] for spec in e.weight_specs()):
            try:
                acts.append(e.activation(name))
            except Exception:
                pass                                     # a tensor the plan fused away
        res.append([z, l] + acts)
        e.close()
        if knob:
            monkeypatch.delenv(knob)
    assert len(res[0]) == len(res[1]) and len(res[0]) > 4
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("arch,shape", [("fcn_skip", (256, 320)), ("fcn", (96, 64)), ("fcn_skip", (1024, 768)), ("fcn_skip", (70, 50))])
def test_bf16_transposed_conv_behind_its_producer(gpu, oracle_mod, monkeypatch, arch, shape):
    """deconv2 (Conv2DTranspose k2 s2) runs on deconv1's accumulators in deconv1's epilogue (FL_DQ): against the engine with the
    two launches (PSEG_NO_DQ) the same products, another float32 summation grouping (k-steps pair the accumulator tiles):
    logits within 2e-3, labels equal except at near-ties, deconv2's tensor within one bf16 ulp step of the unfused one."""
    from pseg_amd import synth
    C = 3
    img = synth.synth_page(11, max(shape[0], 96), max(shape[1], 96), C)[0][:shape[0], :shape[1]].copy()
    Wt = oracle_mod.init_weights(arch, C, seed=21, gain=1.5, bias_scale=0.05)
    res = []
    for knob in (None, "PSEG_NO_DQ"):
        if knob:
            monkeypatch.setenv(knob, "1")
        e = gpu.Engine(arch, C, mode=gpu.MODE_BF16)
        e.set_weights(Wt)
        z, _, l = e.predict(img, want_probs=False)
        d2 = e.activation("conv2d_transpose_1")
        if knob is None:
            with pytest.raises(gpu.PsegError, match="fused"):
                e.activation("conv2d_transpose")
        else:
            assert e.activation("conv2d_transpose").shape[2] == 80
        res.append((z, l, d2))
        e.close()
        if knob:
            monkeypatch.delenv(knob)
    (z1, l1, d1), (z0, l0, d0) = res
    assert d1.shape == d0.shape and np.abs(d1 - d0).max() <= 2 ** -6 * max(1.0, np.abs(d0).max())
    assert np.abs(z1 - z0).max() <= 2e-3 * max(1.0, np.abs(z0).max())
    assert _check_labels(l1, z1, z0)[0] == 0
    z_o = oracle_mod.forward(arch, Wt, img, "f32")
    assert np.abs(z1 - z_o).max() <= TOL * max(1.0, np.abs(z_o).max())


def test_engines_keep_their_own_knob_snapshot(gpu, oracle_mod, monkeypatch):
    """The PSEG_* developer knobs are snapshotted per engine at creation: an engine created earlier keeps working, bit for
    bit, after the environment changed and ANOTHER engine was created under it (launch-time reads used to
    follow the newest snapshot, so engine A's conv1+conv2 launch failed once engine B existed under PSEG_NO_WS), and the
    later engine really runs under its own knobs."""
    Wt = oracle_mod.init_weights("fcn_skip", 3, seed=5, gain=1.5, bias_scale=0.05)
    img = np.random.default_rng(5).integers(0, 256, size=(160, 96), dtype=np.uint8)
    a = gpu.Engine("fcn_skip", 3, mode=gpu.MODE_BF16)
    a.set_weights(Wt)
    za, _, la = a.predict(img, want_probs=False)
    for k in ("PSEG_NO_WS", "PSEG_NO_PERSIST", "PSEG_NO_TAIL2", "PSEG_NO_SKIPLOG"):
        monkeypatch.setenv(k, "1")
    b = gpu.Engine("fcn_skip", 3, mode=gpu.MODE_BF16)
    b.set_weights(Wt)
    zb, _, lb = b.predict(img, want_probs=False)
    act_b = b.activation("conv2d_1")                 # materialised under PSEG_NO_SKIPLOG: engine b reads ITS snapshot
    za2, _, la2 = a.predict(img, want_probs=False)    # engine a: planned with skip-logits fusion, must still launch that way
    assert np.array_equal(za, za2) and np.array_equal(la, la2)
    with pytest.raises(gpu.PsegError):
        a.activation("conv2d_1")                      # ... its conv2 tensor is still fused away
    assert act_b.shape[-1] == 30
    assert np.abs(zb - za).max() <= 2e-2 * max(1.0, float(np.abs(za).max()))
    a.close()
    b.close()


@pytest.mark.parametrize("arch,C,shape", [("fcn_skip", 3, (96, 80)), ("fcn_skip", 6, (130, 67)), ("fcn", 3, (300, 420)), ("fcn_skip", 3, (1056, 1000)),
                                          ("fcn_skip", 3, (33, 1)), ("fcn", 6, (1, 37)), ("fcn_skip", 3, (416, 352)), ("fcn", 3, (700, 1180))])
def test_bf16_two_team_streamed_weights_kernel_is_bit_identical(gpu, monkeypatch, arch, C, shape):
    """conv_sp2_kernel (conv5, conv6, deconv3: two compute teams of four waves on ONE weight ring, a tile pair per workgroup trip,
    four block slots, tiles of 8 x 32 or 8 x 24) against conv_mfma_kernel (PSEG_NO_SP) and against conv_sp_kernel (PSEG_SP2=0 +
    PSEG_SP_ALL): every wave keeps the packing, k order and start value -> the same bits in the logits, the labels and every tensor
    these layers write.  PSEG_SP2=24 / 32 force the kernel with that tile width on every page size: one-tile pages (team 1 gets the
    virtual tile of an odd tile count), ragged edges, several pairs per workgroup, six classes, a fused pool on the narrow tile."""
    from pseg_amd import synth
    img = synth.synth_page(9, shape[0], shape[1], C)[0] if min(shape) >= 64 else np.random.default_rng(9).integers(0, 256, shape, dtype=np.uint8)
    res = []
    for env in ({"PSEG_SP2": "24", "PSEG_SP_CHECK": "1"}, {"PSEG_SP2": "32", "PSEG_SP_CHECK": "1"}, {"PSEG_SP2": "0", "PSEG_SP_ALL": "1", "PSEG_SP_CHECK": "1"},
                {"PSEG_NO_SP": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        e = gpu.Engine(arch, C, mode=gpu.MODE_BF16)
        e.set_weights(synth.glorot_weights(e.weight_specs(), seed=42, gain=1.5, bias_scale=0.05))
        z, _, l = e.predict(img, want_probs=False)
        z2, _, l2 = e.predict(img, want_probs=False)
        assert np.array_equal(z, z2) and np.array_equal(l, l2)
        acts = [z, l]
        for name in ("conv2d_4", "max_pooling2d_2", "conv2d_transpose_2"):
            try:
                acts.append(e.activation(name))
            except Exception:
                acts.append(None)
        res.append(acts)
        e.close()
        for k in env:
            monkeypatch.delenv(k)
    for other in res[1:]:
        for a, b in zip(res[0], other):
            assert (a is None) == (b is None)
            if a is not None:
                assert np.array_equal(a, b)
    assert np.abs(res[0][0]).max() > 0


def test_bf16_two_team_kernel_in_page_units(gpu, monkeypatch):
    """A unit of pages through conv_sp2_kernel (a tile index carries the page; five pages of 5 x 9 narrow tiles = an odd tile
    count) equals the pages one by one, with the kernel (PSEG_SP2=24) and without it (PSEG_SP2=0)."""
    import ctypes
    from pseg_amd import synth
    H, W, n = 160, 200, 5
    pages = np.stack([synth.synth_page(20 + i, H, W, 3)[0] for i in range(n)])
    hip = ctypes.CDLL("libamdhip64.so")
    d_in, d_out = ctypes.c_void_p(), ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(d_in), pages.size) == 0 and hip.hipMalloc(ctypes.byref(d_out), pages.size) == 0
    assert hip.hipMemcpy(d_in, pages.ctypes.data_as(ctypes.c_void_p), pages.size, 1) == 0
    outs = {}
    for sw in ("24", "0"):
        monkeypatch.setenv("PSEG_SP2", sw)
        monkeypatch.setenv("PSEG_SP_CHECK", "1")
        e = gpu.Engine("fcn_skip", 3, mode=gpu.MODE_BF16)
        e.set_weights(synth.glorot_weights(e.weight_specs(), seed=42, gain=1.5, bias_scale=0.05))
        one = np.stack([e.predict(pages[i], want_logits=False, want_probs=False)[2] for i in range(n)]).astype(np.uint8)
        e.predict(pages[0], want_logits=False, want_probs=False)          # (plans exist: the next call may travel as a unit)
        e.predict_pages_device(d_in.value, n, H, W, d_labels_u8=d_out.value)
        e.status()
        got = np.empty_like(pages)
        assert hip.hipMemcpy(got.ctypes.data_as(ctypes.c_void_p), d_out, pages.size, 2) == 0
        assert np.array_equal(got, one), sw
        outs[sw] = got
        e.close()
    assert np.array_equal(outs["24"], outs["0"])
    hip.hipFree(d_in); hip.hipFree(d_out)


def test_plan_switch_in_the_environment_does_not_reach_a_product_engine(gpu, oracle_mod, monkeypatch):
    """the property itself, on the product path.  PSEG_NO_SKIPLOG is a plan switch (not in pseg_env_knobs()): set
    in the process environment it must NOT change an engine created through pseg_create / pseg_create_ex / an empty plan -- its
    conv2 tensor stays fused away (skip-logits fusion) -- while the same switch passed as a plan does materialise the tensor;
    a LISTED knob (PSEG_NO_WS) in the environment is accepted on the same path (labels equal up to near-ties)."""
    import ctypes
    from pseg_amd import engine as E
    Wt = oracle_mod.init_weights("fcn_skip", 3, seed=5, gain=1.5, bias_scale=0.05)
    img = np.random.default_rng(6).integers(0, 256, size=(96, 160), dtype=np.uint8)
    assert "PSEG_NO_SKIPLOG" not in E.lib().pseg_env_knobs().decode().split("\n")
    monkeypatch.setenv("PSEG_NO_SKIPLOG", "1")
    a = gpu.Engine("fcn_skip", 3, mode=gpu.MODE_BF16, plan="")                  # product path: pseg_create_plan(..., "") = pseg_create_ex
    a.set_weights(Wt)
    la = a.predict(img, want_logits=False, want_probs=False)[2]
    with pytest.raises(gpu.PsegError):
        a.activation("conv2d_1")                                                # still fused: the environment did not reach the plan
    # ... and through the bare C entry
    h = ctypes.c_void_p()
    E._check(E.lib().pseg_create(E.ARCH_IDS["fcn_skip"], 3, 1, 0, gpu.MODE_BF16, ctypes.byref(h)))
    c = gpu.Engine.__new__(gpu.Engine)
    c._h, c.arch, c.n_classes, c.in_channels, c.device, c.mode, c.batch_norm = h, "fcn_skip", 3, 1, 0, gpu.MODE_BF16, False
    c.set_weights(Wt)
    lc = c.predict(img, want_logits=False, want_probs=False)[2]
    with pytest.raises(gpu.PsegError):
        c.activation("conv2d_1")
    b = gpu.Engine("fcn_skip", 3, mode=gpu.MODE_BF16, plan={"PSEG_NO_SKIPLOG": "1"})
    b.set_weights(Wt)
    b.predict(img, want_logits=False, want_probs=False)
    assert b.activation("conv2d_1").shape[-1] == 30                             # the plan switch does what the environment could not
    monkeypatch.delenv("PSEG_NO_SKIPLOG")
    monkeypatch.setenv("PSEG_NO_WS", "1")                                       # a listed knob: read from the environment at creation
    d = gpu.Engine("fcn_skip", 3, mode=gpu.MODE_BF16, plan="")
    d.set_weights(Wt)
    ld = d.predict(img, want_logits=False, want_probs=False)[2]
    assert np.array_equal(la, lc)
    assert (la != ld).mean() < 0.02       # (the fused instance sums conv2's products in another order: a near-tie may flip, nothing more)
    for e in (a, b, c, d):
        e.close()


_GIVE_UP_SCRIPT = r"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.join(os.environ["PSEG_ROOT"], "page-segmentation_amd"))
import pseg_amd
from pseg_amd import synth
from pseg_amd.engine import PsegError
assert "PSEG_SP_CHECK" not in os.environ
img = synth.synth_page(9, 300, 420, 3)[0]
e = pseg_amd.Engine("fcn_skip", 3, mode=pseg_amd.MODE_BF16)
e.set_weights(synth.glorot_weights(e.weight_specs(), seed=42, gain=1.5, bias_scale=0.05))
good = e.predict(img, want_logits=False, want_probs=False)[2]
out = {}
os.environ["PSEG_SP_DBG"] = "8"            # diagnostic build: the weight loaders of conv_sp_kernel stop, every wait gives up after 64 polls
for name, call in (("predict", lambda: e.predict(img, want_logits=False, want_probs=False)),
                   ("predict_batch", lambda: e.predict_batch([img, img])),
                   ("predict_chain", lambda: e.predict_chain(img))):
    try:
        call()
        out[name] = "no error"
    except PsegError as ex:
        out[name] = str(ex)
# an asynchronous entry cannot tell: pseg_engine_status does, once
import ctypes
from pseg_amd.engine import lib
L = lib()
d_img, d_lab = ctypes.c_void_p(), ctypes.c_void_p()
hip = ctypes.CDLL("libamdhip64.so")
assert hip.hipMalloc(ctypes.byref(d_img), img.size) == 0 and hip.hipMalloc(ctypes.byref(d_lab), img.size) == 0
assert hip.hipMemcpy(d_img, img.ctypes.data_as(ctypes.c_void_p), img.size, 1) == 0
e.predict_device(d_img.value, 300, 420, d_labels_u8=d_lab.value)
try:
    e.status()
    out["status"] = "no error"
except PsegError as ex:
    out["status"] = str(ex)
try:
    e.status()
    out["status_again"] = "no error"
except PsegError as ex:
    out["status_again"] = str(ex)
del os.environ["PSEG_SP_DBG"]
again = e.predict(img, want_logits=False, want_probs=False)[2]     # the record was cleared by the report: the engine works again
out["recovered"] = bool(np.array_equal(good, again))
print("RESULT " + json.dumps(out))
"""


def test_bf16_streamed_weights_kernel_give_up_is_an_error_in_the_default_path(gpu):
    """A counter wait of conv_sp_kernel that gives up (bounded polling: a protocol bug must not hang the GPU) leaves a record in
    device memory; EVERY host-synchronous entry reads it before returning and fails with PSEG_EHIP -- without PSEG_SP_CHECK, which
    only adds a check after each launch -- and pseg_engine_status reports it for the asynchronous entries, once (report = clear).
    The give-up is forced in the diagnostic build (libpseg_diag.so, PSEG_SP_DBG=8: the weight loaders stop after their first
    groups); the release library does not contain the switch."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    diag = os.path.join(root, "page-segmentation_amd", "csrc", "libpseg_diag.so")
    assert os.path.exists(diag), "libpseg_diag.so is not built (__graft_entry__.build() builds it)"
    env = {k: v for k, v in os.environ.items() if not k.startswith("PSEG_")}
    env.update({"PSEG_LIB": diag, "PSEG_ROOT": root})
    r = subprocess.run([sys.executable, "-c", _GIVE_UP_SCRIPT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    for name in ("predict", "predict_batch", "predict_chain", "status"):
        assert "a counter wait gave up" in out[name] and "conv2d" in out[name], (name, out[name])
    assert out["status_again"] == "no error"
    assert out["recovered"] is True
