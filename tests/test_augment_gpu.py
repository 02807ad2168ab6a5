"""Augmentation (lib/data_generator.py, lib/network.py:109-125,149-161) on the GPU: the affine warp against the
installed scipy.ndimage.affine_transform (float32 rounding apart), the parameter stream of the generator, and
the augmented sample stream of Network.create_dataset_inputs."""
import numpy as np
import pytest

from oracle import augment as A

pytestmark = pytest.mark.gpu


def _params(h, w, theta, tx, ty, zx, zy, shear=0.0):
    from ocr4all_pixel_classifier.lib.data_generator import ImageDataGeneratorCustom as G
    return G.affine_matrix({'theta': theta, 'tx': tx, 'ty': ty, 'shear': shear, 'zx': zx, 'zy': zy}, h, w)


@pytest.mark.parametrize("shape", [(64, 96), (33, 50), (257, 131), (7, 5)])
@pytest.mark.parametrize("order", [0, 3])
def test_affine_warp_matches_scipy(gpu, shape, order):
    from pseg_amd import engine as E
    rng = np.random.default_rng(shape[0] * 10 + order)
    x = (rng.random(shape) * 255).astype(np.float32)
    if order == 0:
        x = np.round(x / 50)                               # label-like
    for theta, tx, ty, zx, zy in [(2.5, 1.6, -2.4, 0.95, 1.05), (-1.3, 0.0, 0.0, 1.0, 1.0), (0.0, 3.0, 2.0, 1.0, 1.0),
                                  (40.0, 5.0, -7.0, 0.7, 1.4)]:
        m, off = _params(shape[0], shape[1], theta, tx, ty, zx, zy)
        got = E.affine_warp(x, m, off, order)
        want = A.affine_transform(x, m, off, order)
        assert got.dtype == np.float32 and got.shape == want.shape
        if order == 0:
            assert (got != want).mean() <= 0.002            # exact .5 coordinate ties only
        else:
            assert np.abs(got - want).max() <= 2e-3 * 255    # B-spline overshoot scale; typically 1e-5


@pytest.mark.parametrize("shape", [(64, 96), (33, 50), (7, 5), (1, 9)])
@pytest.mark.parametrize("order", [0, 3])
def test_affine_warp_constant_fill_matches_scipy(gpu, shape, order):
    """fill_mode='constant' with a cval (lib/trainer.py:23-28 image_/binary_/mask_fill_mode and *_cval) against the installed
    scipy.ndimage.affine_transform(mode='constant'): the fill value wherever the source coordinate leaves the plane, the
    unpadded spline prefilter inside; 'reflect' / 'wrap' raise."""
    from scipy import ndimage
    from pseg_amd import engine as E
    rng = np.random.default_rng(shape[0] * 10 + order + 1)
    x = (rng.random(shape) * 255).astype(np.float32)
    if order == 0:
        x = np.round(x / 50)
    for cval in (0.0, 255.0, 3.0):
        for theta, tx, ty, zx, zy in [(2.5, 1.6, -2.4, 0.95, 1.05), (0.0, 3.0, 2.0, 1.0, 1.0), (40.0, 5.0, -7.0, 0.7, 1.4)]:
            m, off = _params(shape[0], shape[1], theta, tx, ty, zx, zy)
            got = E.affine_warp(x, m, off, order, fill_mode="constant", cval=cval)
            want = ndimage.affine_transform(x.astype(np.float64), m, off, order=order, mode="constant", cval=cval).astype(np.float32)
            assert got.dtype == np.float32 and got.shape == want.shape
            if order == 0:
                assert (got != want).mean() <= 0.01             # .5 ties, and coordinates within rounding of the plane's edge
            else:
                edge = (got == np.float32(cval)) != (want == np.float32(cval))      # coordinates within float64 rounding of the edge
                assert edge.mean() <= 0.01
                assert np.abs(got - want)[~edge].max(initial=0.0) <= 2e-3 * 255
    with pytest.raises(E.PsegError):
        E.affine_warp(x, np.eye(2), np.zeros(2), order, fill_mode="mirror")        # not a keras-preprocessing mode
    from ocr4all_pixel_classifier.lib.data_generator import ImageDataGeneratorCustom as G
    g = G(rotation_range=5, fill_mode="constant", cval=7.0, interpolation_order=order)
    img = np.stack([x, x[::-1]], -1)
    p = {'theta': 5.0, 'tx': 2.0, 'ty': -1.0, 'shear': 0.0, 'zx': 0.9, 'zy': 1.1, 'flip_horizontal': False, 'flip_vertical': False}
    out = g.apply_transform(img, p)
    m, off = G.affine_matrix(p, shape[0], shape[1])
    assert np.array_equal(out[..., 1], E.affine_warp(np.ascontiguousarray(img[..., 1]), m, off, order, fill_mode="constant", cval=7.0))
    with pytest.raises(Exception):
        G(fill_mode="grid-wrap")


@pytest.mark.parametrize("shape", [(64, 96), (33, 50), (7, 5), (1, 9), (200, 31)])
@pytest.mark.parametrize("order", [0, 3])
@pytest.mark.parametrize("mode", ["reflect", "wrap"])
def test_affine_warp_reflect_and_wrap_fill_match_scipy(gpu, shape, order, mode):
    """fill_mode 'reflect' / 'wrap' (the other two values lib/trainer.py:23-25's *_fill_mode fields may carry) against the
    installed scipy.ndimage.affine_transform: coordinates mapped by the mode (reflect: half-sample symmetric; wrap: scipy's
    period n - 1), the prefilter with the mode's boundary (reflect exact; wrap: mirror), tap indices mapped by the prefilter's
    boundary.  Large shifts and rotations so that several periods are crossed."""
    from pseg_amd import engine as E
    rng = np.random.default_rng(shape[0] * 10 + order + (7 if mode == "wrap" else 3))
    x = (rng.random(shape) * 255).astype(np.float32)
    if order == 0:
        x = np.round(x / 50)
    for theta, tx, ty, zx, zy in [(2.5, 1.6, -2.4, 0.95, 1.05), (0.0, 3.0, 2.0, 1.0, 1.0), (40.0, 5.0, -7.0, 0.7, 1.4), (-170.0, 90.5, -140.25, 2.5, 0.4)]:
        m, off = _params(shape[0], shape[1], theta, tx, ty, zx, zy)
        got = E.affine_warp(x, m, off, order, fill_mode=mode)
        want = A.affine_transform_mode(x, m, off, order, mode)
        assert got.dtype == np.float32 and got.shape == want.shape
        if order == 0:
            assert (got != want).mean() <= 0.01                 # exact .5 coordinate ties / coordinates within rounding of a period's seam
        else:
            d = np.abs(got - want)
            seam = d > 2e-3 * 255                               # a coordinate within float64 rounding of a seam lands on the other side
            assert seam.mean() <= 0.01 and np.median(d) <= 1e-3
    from ocr4all_pixel_classifier.lib.data_generator import ImageDataGeneratorCustom as G
    from ocr4all_pixel_classifier.lib.trainer import AugmentationSettings
    s = AugmentationSettings(image_fill_mode=mode, binary_fill_mode=mode, mask_fill_mode=mode)
    g = G(**(s.to_image_params() if order == 3 else s.to_mask_params()), data_format='channels_last')
    assert g.fill_mode == mode
    p = {'theta': 5.0, 'tx': 2.0, 'ty': -1.0, 'shear': 0.0, 'zx': 0.9, 'zy': 1.1, 'flip_horizontal': False, 'flip_vertical': False}
    out = g.apply_transform(x[..., None], p)
    m, off = G.affine_matrix(p, shape[0], shape[1])
    assert np.array_equal(out[..., 0], E.affine_warp(x, m, off, order, fill_mode=mode))


@pytest.mark.parametrize("case", ["in_range", "overshoot", "negative", "flat", "rgb"])
def test_brightness_shift_matches_the_pillow_restatement(gpu, case):
    """AugmentationSettings.brightness_range (lib/trainer.py:21): pseg_brightness_shift against oracle.augment.apply_brightness_shift
    (keras-preprocessing 1.1.2's apply_brightness_shift(x, b, scale=False) run through the installed Pillow), bit for bit:
    planes inside [0, 255] (truncation to uint8, truncating blend), planes the cubic warp pushed outside it (stretch to 8 bit and
    back), factors below and above one (the clipped extrapolation branch), a constant plane, three channels."""
    from pseg_amd import engine as E
    rng = np.random.default_rng(len(case))
    H, W = 37, 53
    x = {"in_range": lambda: rng.random((H, W, 1)) * 255, "overshoot": lambda: rng.random((H, W, 1)) * 280 - 10,
         "negative": lambda: rng.random((H, W, 1)) * 200 - 60, "flat": lambda: np.full((H, W, 1), 93.7),
         "rgb": lambda: rng.random((H, W, 3)) * 262 - 3}[case]().astype(np.float32)
    for b in (0.0, 0.37, 0.8, 1.0, 1.2, 1.9, 3.5):
        got = E.brightness_shift(x, b)
        want = A.apply_brightness_shift(x, b)
        assert got.dtype == np.float32 and got.shape == want.shape
        assert np.array_equal(got, want), (case, b, float(np.abs(got - want).max()))


def test_generator_brightness_range_reaches_the_image_only(gpu):
    """The brightness draw is the last of get_random_transform: image, binary and mask generators (lib/network.py:149-153 share one
    seed) still see the same affine parameters, and only the image generator has a 'brightness'; flow() applies it after warp
    and flips."""
    from ocr4all_pixel_classifier.lib.data_generator import ImageDataGeneratorCustom as G
    from ocr4all_pixel_classifier.lib.trainer import AugmentationSettings
    from pseg_amd import engine as E
    s = AugmentationSettings(brightness_range=[0.6, 1.4], horizontal_flip=True)
    gi, gm = G(**s.to_image_params(), data_format='channels_last'), G(**s.to_mask_params(), data_format='channels_last')
    assert gi.brightness_range == [0.6, 1.4] and gm.brightness_range is None
    pi, pm = gi.get_random_transform((64, 96, 1), seed=11), gm.get_random_transform((64, 96, 1), seed=11)
    assert 0.6 <= pi['brightness'] <= 1.4 and pm['brightness'] is None
    assert all(pi[k] == pm[k] for k in pm if k != 'brightness')
    np.random.seed(11)
    rs = np.random.RandomState(11)
    for _ in range(3): rs.uniform(-1, 1)
    rs.uniform(0.95, 1.05, 2); rs.random_sample(); rs.random_sample()
    assert pi['brightness'] == rs.uniform(0.6, 1.4)                    # theta, tx, ty, (zx, zy), two flip draws, then brightness
    x = (np.random.default_rng(2).random((1, 64, 96, 1)) * 255).astype(np.float32)
    out = next(gi.flow(x, seed=5, batch_size=1))[0]
    np.random.seed(5); np.random.permutation(1)
    p = gi.get_random_transform((64, 96, 1))
    no_b = dict(p, brightness=None)
    assert np.array_equal(out, E.brightness_shift(gi.apply_transform(x[0], no_b), p['brightness']))
    with pytest.raises(ValueError):
        G(brightness_range=0.5)


def test_generator_parameter_stream_and_flow(gpu):
    from ocr4all_pixel_classifier.lib.data_generator import ImageDataGeneratorCustom as G
    from ocr4all_pixel_classifier.lib.trainer import AugmentationSettings
    s = AugmentationSettings()
    gi, gb = G(**s.to_image_params(), data_format='channels_last'), G(**s.to_binary_params(), data_format='channels_last')
    assert gi.interpolation_order == 3 and gb.interpolation_order == 0
    # same seed -> same parameters for image, binary and mask generators (lib/network.py:149-153)
    np.random.seed(7)
    np.random.permutation(1)
    p1 = gi.get_random_transform((64, 96, 1))
    np.random.seed(7)
    np.random.permutation(1)
    p2 = gb.get_random_transform((64, 96, 1))
    assert p1 == p2 and abs(p1['theta']) <= 2.5 and abs(p1['tx']) <= 0.025 * 64 and 0.95 <= p1['zx'] <= 1.05
    # the draw order of keras-preprocessing: theta, tx, ty, (shear), zoom pair, two flip draws
    np.random.seed(7)
    exp_theta = np.random.uniform(-2.5, 2.5)
    exp_tx = np.random.uniform(-0.025, 0.025) * 64
    exp_ty = np.random.uniform(-0.025, 0.025) * 96
    exp_z = np.random.uniform(0.95, 1.05, 2)
    assert (p1['theta'], p1['tx'], p1['ty'], p1['zx'], p1['zy']) == (exp_theta, exp_tx, exp_ty, exp_z[0], exp_z[1])
    rng = np.random.default_rng(1)
    img = (rng.random((1, 64, 96, 1)) * 255).astype(np.uint8)
    a = next(gi.flow(img, seed=11, batch_size=1))
    b = next(gi.flow(img, seed=11, batch_size=1))
    c = next(gi.flow(img, seed=12, batch_size=1))
    assert a.shape == (1, 64, 96, 1) and a.dtype == np.float32 and np.array_equal(a, b) and not np.array_equal(a, c)
    m, off = G.affine_matrix(gi.get_random_transform((64, 96, 1), seed=11), 64, 96)
    assert np.abs(a[0, ..., 0] - A.affine_transform(img[0, ..., 0].astype(np.float32), m, off, 3)).max() < 0.5
    with pytest.raises(Exception):
        G(fill_mode='reflect')


def test_create_dataset_inputs_with_augmentation(gpu):
    from pseg_amd import synth
    from ocr4all_pixel_classifier.lib.network import Network
    from ocr4all_pixel_classifier.lib.dataset import Dataset, SingleData
    from ocr4all_pixel_classifier.lib.colors import ColorMap
    from ocr4all_pixel_classifier.lib.trainer import AugmentationSettings
    img, binary, mask = synth.synth_page(3, 96, 128, 3)
    ds = Dataset([SingleData(image=img, binary=binary, mask=mask, original_shape=img.shape)], ColorMap({}))
    net = Network("train", n_classes=3, exact=True)
    gen = net.create_dataset_inputs(ds, data_augmentation=True, data_augmentation_settings=AugmentationSettings())
    (x1, y1), (x2, y2) = next(gen), next(gen)
    assert x1['input_1'].shape == (1, 96, 128, 1) and x1['input_2'].shape == (1, 96, 128, 1) and y1['logits'].shape == (1, 96, 128, 1)
    assert x1['input_1'].max() <= 1.3 and x1['input_1'].min() >= -0.3             # / 255, cubic-spline overshoot kept (scipy too)
    assert set(np.unique(y1['logits'])) <= {0.0, 1.0, 2.0}                          # order-0 warp keeps label values
    assert not np.array_equal(x1['input_1'], x2['input_1'])                         # seed += 1 per sample
    # mask and image moved together: the label map still agrees with the un-augmented one on most pixels
    assert (y1['logits'][0, ..., 0] == mask).mean() > 0.9


def test_float_page_training_entry_and_augmented_training(gpu, oracle_mod, tmp_path):
    """pseg_train_forward_backward_f32 on an un-augmented page equals the uint8 entry (x / 255.0f either way; atomics order apart), and Network.train_dataset with data_augmentation=True learns."""
    from pseg_amd import synth
    from ocr4all_pixel_classifier.lib.trainer import Trainer, TrainSettings
    from ocr4all_pixel_classifier.lib.dataset import Dataset, SingleData
    from ocr4all_pixel_classifier.lib.colors import ColorMap
    from ocr4all_pixel_classifier.lib.metrics import Monitor
    img, binary, mask = synth.synth_page(2, 96, 112, 3)
    Wt = oracle_mod.init_weights("fcn_skip", 3, seed=3, gain=1.0, bias_scale=0.02)
    e = gpu.Engine("fcn_skip", 3, mode=gpu.MODE_F32_EXACT)
    e.set_weights(Wt)
    e.train_init()
    a = e.train_forward_backward(img, mask)
    ga = {k: v.copy() for k, v in e.gradients().items()}
    b = e.train_forward_backward_float(img.astype(np.float32), mask)
    gb = e.gradients()
    assert np.allclose(a, b, rtol=1e-6, atol=0)                       # same forward; reductions use float atomics
    assert all(np.allclose(ga[k], gb[k], rtol=1e-4, atol=1e-6 * np.abs(ga[k]).max()) for k in ga)
    e.close()
    np.random.seed(0)
    cm = ColorMap({})

    def ds(seeds):
        out = []
        for s in seeds:
            i, bi, m = synth.synth_page(s, 96, 96, 3)
            out.append(SingleData(image=i, binary=bi, mask=m, original_shape=i.shape))
        return Dataset(out, cm)
    settings = TrainSettings(n_epoch=6, n_classes=3, l_rate=2e-3, train_data=ds([0, 1, 2, 3]), validation_data=ds([4]),
                             display=1, output_dir=str(tmp_path), threads=1, monitor=Monitor.VAL_LOSS,
                             data_augmentation=True)
    hist = Trainer(settings).train()
    assert len(hist["loss"]) == 6 and np.isfinite(hist["loss"]).all()
    assert np.mean(hist["loss"][-2:]) < hist["loss"][0]
