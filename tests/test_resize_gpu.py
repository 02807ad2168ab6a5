"""Line-height normalisation kernels (nearest / Gaussian anti-aliasing / bicubic / prepare_images)
vs oracle/resize.py: BIT-EXACT (float64 planes and uint8 outputs, same anti-aliasing kernels), and
against the scikit-image goldens with the documented integer-boundary tolerance."""
import os

import numpy as np
import pytest

from oracle import resize as R

pytestmark = pytest.mark.gpu

G = {k.replace("__", "/"): v for k, v in
     np.load(os.path.join(os.path.dirname(__file__), "golden", "resize_vectors.npz"), allow_pickle=False).items()}


def _page(seed, H, W):
    rng = np.random.default_rng(seed)
    img = np.clip(rng.normal(215, 12, (H, W)), 0, 255)
    for y in range(6, H - 20, 24):
        x = 5
        while x < W - 16:
            w, h = int(rng.integers(4, 14)), int(rng.integers(8, 18))
            img[y:y + h, x:x + w] = np.clip(rng.normal(45, 18, (h, min(w, W - x))), 0, 255)
            x += w + int(rng.integers(2, 7))
    img = img.astype(np.uint8)
    return img, np.where(img > 127, 255, 0).astype(np.uint8)


@pytest.mark.parametrize("name", ("down", "up", "down_mw", "flat", "aniso", "twoval"))
def test_prepare_images_golden_inputs(gpu, name):
    from pseg_amd import engine as E
    tgt, lh, mw = (int(v) for v in G[name + "/params"])
    mw = None if mw < 0 else mw
    image, binary = G[name + "/image"], G[name + "/binary"]
    want = R.prepare_images(image, binary, tgt, lh, mw)
    got = E.prepare_images(image, binary, tgt / lh, mw, want_stage1=True)
    for g, w, what in zip(got, want, ("img", "bin", "orig_bin", "stage1")):
        assert g.dtype == w.dtype and g.shape == w.shape, what
        assert np.array_equal(g, w), what                       # bit-exact, float64 stage included
    # and against scikit-image itself (tolerances: tests/test_resize_oracle.py)
    assert np.abs(got[3] - G[name + "/stage1"]).max() < 1e-9
    d = np.abs(got[0].astype(int) - G[name + "/out_img"].astype(int))
    assert d.max() <= 1 and (d != 0).mean() <= 0.005


@pytest.mark.parametrize("H,W,tgt,lh,mw", [
    (301, 517, 6, 23, None), (257, 193, 6, 5, None), (480, 1300, 8, 21, 300), (64, 64, 6, 6, None),
    (33, 1, 6, 14, None), (1, 57, 6, 14, None), (7, 5, 2, 28, None), (123, 77, 6, 7, 40)])
def test_prepare_images_random(gpu, H, W, tgt, lh, mw):
    from pseg_amd import engine as E
    image, binary = _page(H * 7 + W, H, W)
    if min(R.rescale_shape((H, W), tgt / lh)) < 1:
        with pytest.raises(Exception):
            E.prepare_images(image, binary, tgt / lh, mw)
        return
    want = R.prepare_images(image, binary, tgt, lh, mw)
    got = E.prepare_images(image, binary, tgt / lh, mw, want_stage1=True)
    for g, w in zip(got, want):
        assert g.shape == w.shape and np.array_equal(g, w)


def test_identity_scale_and_binary_conventions(gpu):
    from pseg_amd import engine as E
    image, binary = _page(5, 80, 60)
    img, b, orig = E.prepare_images(image, binary, 1.0)
    assert np.array_equal(img, ((1.0 - image.astype(np.float64) / 255) * 255).astype(np.uint8))    # lib/dataset.py:137,145
    assert np.array_equal(b, (binary == 0).astype(np.uint8)) and np.array_equal(orig, b)
    img01, b01, orig01 = E.prepare_images(image, binary // 255, 1.0)           # 0/1 binaries: not divided by 255
    assert np.array_equal(b01, b) and np.array_equal(img01, img) and np.array_equal(orig01, orig)
    odd = binary.copy()
    odd[0, 0] = 128                                                             # 1 - 128/255 truncates to 0
    assert E.prepare_images(image, odd, 1.0)[1][0, 0] == 0


@pytest.mark.parametrize("dtype", (np.uint8, np.int64, np.float32, np.float64, np.uint16))
@pytest.mark.parametrize("shape,out", [((37, 53), (91, 120)), ((37, 53), (13, 22)), ((5, 4), (5, 4)), ((1, 9), (3, 2))])
def test_resize_nearest(gpu, dtype, shape, out):
    from pseg_amd import engine as E
    rng = np.random.default_rng(3)
    a = (rng.random(shape) * 200).astype(dtype)
    got = E.resize_nearest(a, out)
    assert got.dtype == a.dtype and np.array_equal(got.astype(np.float64), R.resize_nearest(a, out))
    rgb = (rng.random(shape + (3,)) * 255).astype(np.uint8)                    # 3-byte pixels (colour masks)
    g3 = E.resize_nearest(rgb, out)
    assert all(np.array_equal(g3[..., c].astype(np.float64), R.resize_nearest(rgb[..., c], out)) for c in range(3))


def test_preserving_resize_api_and_goldens(gpu):
    from ocr4all_pixel_classifier.lib.util import preserving_resize
    for nm in ("pr_up", "pr_down"):
        out = preserving_resize(G[nm + "/in"], G[nm + "/out"].shape)
        assert out.dtype == np.float64 and np.array_equal(out, G[nm + "/out"])
    a = np.arange(12).reshape(3, 4)
    assert np.array_equal(preserving_resize(a, (6, 8)), np.repeat(np.repeat(a, 2, 0), 2, 1))
    assert np.array_equal(preserving_resize(np.array([[True, False]]), (2, 4)), [[1, 1, 0, 0]] * 2)


def test_scale_image_float_input_and_two_valued(gpu):
    from pseg_amd import engine as E
    rng = np.random.default_rng(11)
    f = rng.random((90, 140))
    assert np.array_equal(E.scale_image(f, (31, 47)), R.scale_image(f, (31, 47)))
    two = (rng.random((90, 140)) > 0.5).astype(np.float64)                      # <= 2 values: no anti-aliasing
    got = E.scale_image(two, (31, 47))
    assert np.array_equal(got, R.resize_bicubic(two, (31, 47))) and got.min() >= 0 and got.max() <= 1
    u = (f * 255).astype(np.uint8)
    assert np.array_equal(E.scale_image(u, (200, 150)), R.scale_image(u, (200, 150)))   # upscale: sigma 0


def test_loader_end_to_end_with_rescale(gpu, tmp_path):
    """DatasetLoader (lib/dataset.py:160-208) through the GPU kernels at line_height != target."""
    import json
    from PIL import Image
    from ocr4all_pixel_classifier.lib.dataset import DatasetLoader
    from ocr4all_pixel_classifier.lib.colors import ColorMap
    gray, _ = _page(9, 240, 180)
    mask_rgb = np.zeros((240, 180, 3), np.uint8)
    mask_rgb[30:120, 20:100] = (255, 0, 0)
    Image.fromarray(gray).save(tmp_path / "p.png")
    Image.fromarray(mask_rgb).save(tmp_path / "m.png")
    cm = ColorMap({"(0, 0, 0)": [0, "bg"], "(255, 0, 0)": [1, "text"]})
    js = {"train": [{"binary_path": str(tmp_path / "p.png"), "image_path": str(tmp_path / "p.png"),
                     "mask_path": str(tmp_path / "m.png"), "line_height_px": 15}], "test": [], "eval": []}
    (tmp_path / "d.json").write_text(json.dumps(js))
    d = DatasetLoader(6, cm).load_data_from_json([str(tmp_path / "d.json")], "train").data[0]
    binary = np.where(gray > 127, 255, 0).astype(np.uint8)
    w_img, w_bin, w_orig, _ = R.prepare_images(gray, binary, 6, 15)
    assert d.original_shape == (240, 180) and d.image.shape == (96, 72)
    assert np.array_equal(d.image, w_img) and np.array_equal(d.binary, w_bin) and np.array_equal(d.orig_binary, w_orig)
    lab = np.zeros((240, 180), np.uint8)
    lab[30:120, 20:100] = 1
    assert np.array_equal(d.mask, R.resize_nearest(lab, (96, 72)).astype(np.uint8))


def test_full_size_scan_properties(gpu):
    """A 3508x2480 (A4 at 300 dpi) scan: size-independent properties instead of the slow oracle."""
    from pseg_amd import engine as E
    image, binary = _page(21, 3508, 2480)
    img, b, orig = E.prepare_images(image, binary, 6 / 25)
    assert img.shape == b.shape == E.rescale_shape((3508, 2480), 6 / 25) and orig.shape == image.shape
    assert set(np.unique(b)) <= {0, 1} and np.array_equal(orig, (binary == 0).astype(np.uint8))
    # the mean intensity survives anti-aliased resampling (inverted scale), ink fraction roughly too
    assert abs(img.mean() - (255 - image.astype(np.float64)).mean()) < 2.0
    assert abs(b.mean() - orig.mean()) < 0.02
    again = E.prepare_images(image, binary, 6 / 25)
    assert all(np.array_equal(x, y) for x, y in zip((img, b, orig), again))     # deterministic
    # mirror equivariance (the warp grid is symmetric; float rounding may move integer-boundary pixels by one)
    fimg, fb, _ = E.prepare_images(image[:, ::-1], binary[:, ::-1], 6 / 25)
    d = np.abs(fimg[:, ::-1].astype(int) - img.astype(int))
    assert d.max() <= 1 and (d != 0).mean() < 0.005 and (fb[:, ::-1] != b).mean() < 0.005
