"""oracle/resize.py against scikit-image 0.18.3 / scipy 1.7.1 outputs (tests/golden/resize_vectors.npz,
made by tests/golden/make_resize_golden.py under the container's conda interpreter).

What can and cannot be bit-identical (documented in oracle/resize.py): scikit-image estimates the
resize affine by least squares, so its coefficients carry ~1e-16 LAPACK-dependent noise, and NumPy
versions differ in the last bit of exp(); both perturb the float64 bicubic value by ~1e-12.  Shapes,
the nearest gathers away from exact .5 ties and the float64 stage-1 image (to 1e-9) must agree; the
final uint8 image may differ by one unit on the few pixels whose value sits on an integer boundary."""
import os

import numpy as np
import pytest

from oracle import resize as R

G = {k.replace("__", "/"): v for k, v in
     np.load(os.path.join(os.path.dirname(__file__), "golden", "resize_vectors.npz"), allow_pickle=False).items()}
CASES = ("down", "up", "down_mw", "flat", "aniso", "twoval")


@pytest.mark.parametrize("name", CASES)
def test_prepare_images_matches_skimage(name):
    tgt, lh, mw = (int(v) for v in G[name + "/params"])
    img, b, orig, st1 = R.prepare_images(G[name + "/image"], G[name + "/binary"], tgt, lh, None if mw < 0 else mw)
    e_img, e_bin = G[name + "/out_img"], G[name + "/out_bin"]
    assert img.shape == e_img.shape and b.shape == e_bin.shape
    assert np.array_equal(orig, G[name + "/out_orig_bin"])
    assert np.abs(st1 - G[name + "/stage1"]).max() < 1e-9
    d = np.abs(img.astype(int) - e_img.astype(int))
    assert d.max() <= 1 and (d != 0).mean() <= 0.005            # integer-boundary pixels only
    # binary: differences only where the nearest coordinate is an exact .5 tie (scale p/q coincidences)
    bad = np.argwhere(b != e_bin)
    if len(bad):
        assert mw < 0
        H0, W0 = G[name + "/image"].shape
        fy, fx = H0 / b.shape[0], W0 / b.shape[1]
        for r, c in bad:
            ty, tx = fy * r + (fy / 2 - 0.5), fx * c + (fx / 2 - 0.5)
            assert min(abs(ty % 1 - 0.5), abs(tx % 1 - 0.5)) < 1e-9
        assert len(bad) <= 0.01 * b.size


@pytest.mark.parametrize("name", ("pr_up", "pr_down"))
def test_preserving_resize_matches_skimage(name):
    out = R.resize_nearest(G[name + "/in"], G[name + "/out"].shape)
    assert out.dtype == np.float64 and np.array_equal(out, G[name + "/out"])


def test_gaussian_kernels_match_scipy_to_a_few_ulp():
    for i in range(4):
        v = G["gauss/%d" % i]
        w, r = R.gaussian_kernel(float(v[0]))
        assert r == int(v[1]) and len(w) == 2 * r + 1
        assert np.all(np.abs(w - v[2:]) <= 4 * np.spacing(v[2:]))  # NumPy 1.26 vs 2.x exp: last bits


def test_rescale_shape_rounds_half_to_even():
    assert R.rescale_shape((5, 7), 0.5) == (2, 4)
    assert R.rescale_shape((61, 83), 6 / 23) == (16, 22)


def test_filter_keeps_uint8_between_passes_and_mirrors():
    a = np.zeros((5, 9), np.uint8)
    a[2, 4] = 255
    f = R.gaussian_filter(a, (1.0, 1.0))
    assert f.dtype == np.uint8 and f[2, 4] == int(int(255 * R.gaussian_kernel(1.0)[0][4]) * R.gaussian_kernel(1.0)[0][4])
    one = np.full((1, 7), 9, np.uint8)                               # a single row mirrors onto itself
    g = R.gaussian_filter(one, (1.5, 0.0))
    assert g.shape == one.shape and g.dtype == np.uint8 and set(np.unique(g)) <= {8, 9}
