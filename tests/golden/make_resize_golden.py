"""Golden vectors for the line-height normalisation path.

Run with the container's conda interpreter, the only one that has scikit-image:
    /opt/conda/bin/python3.9 tests/golden/make_resize_golden.py
scikit-image 0.18.3 / scipy 1.7.1 stand in for the reference's pinned scikit-image 0.17.2
(requirements.txt:131; same resize -> gaussian_filter -> warp -> _warp_fast route).  The reference's
lib/dataset.py cannot be imported (needs the absent `ocr4all` package), so the three library calls
are made here with exactly the keyword arguments of lib/dataset.py:115-119,123-128 and
lib/util.py:28-29, composed in the order of lib/dataset.py:131-150.
Writes tests/golden/resize_vectors.npz (inputs + expected outputs, plain arrays).
"""
import os

import numpy as np
from scipy.ndimage.filters import _gaussian_kernel1d
from skimage.transform import rescale, resize


def lib_scale_binary(b, scale):
    return rescale(b, scale, order=0, anti_aliasing=False, preserve_range=True, multichannel=False)


def lib_scale_image(img, shape):
    return resize(img, shape, order=3, anti_aliasing=len(np.unique(img)) > 2, preserve_range=True)


def lib_prepare(image, binary, target, lh, max_width=None):
    scale = target / lh
    ob = binary / 255 if np.max(binary) > 1 else binary
    b = 1.0 - lib_scale_binary(ob, scale)
    stage1 = lib_scale_image(image, b.shape)
    im = 1.0 - stage1 / 255
    if max_width is not None:
        n = max_width / b.shape[1]
        if n < 1.0:
            b = lib_scale_binary(b, n)
            im = lib_scale_image(im, b.shape)
    return (im * 255).astype(np.uint8), b.astype(np.uint8), (1 - ob).astype(np.uint8), stage1


def page(seed, H, W, flat=False):
    rng = np.random.default_rng(seed)
    img = np.clip(rng.normal(225, 8, (H, W)), 0, 255)
    if flat:
        img[:] = 200.0
    for y in range(4, H - 12, 14):
        x = 3
        while x < W - 12:
            w, h = int(rng.integers(3, 9)), int(rng.integers(5, 11))
            img[y:y + h, x:x + w] = np.clip(rng.normal(40, 15, (h, min(w, W - x))), 0, 255) if not flat else 37.0
            x += w + int(rng.integers(2, 5))
    img = img.astype(np.uint8)
    if flat:
        img[H // 2:, W // 2:] = 118
    binary = np.where(img > 127, 255, 0).astype(np.uint8)
    return img, binary


def main():
    out = {}
    cases = [
        ("down", 1, 61, 83, 6, 23, None, False),      # typical: downscale with anti-aliasing
        ("up", 2, 40, 52, 6, 5, None, False),         # upscale: sigma = 0
        ("down_mw", 3, 57, 131, 7, 17, 37, False),    # second stage (max_width) on the float image
        ("flat", 4, 64, 96, 6, 19, None, True),       # flat areas: truncation-sensitive
        ("aniso", 5, 75, 49, 6, 13, None, False),
    ]
    for name, seed, H, W, tgt, lh, mw, flat in cases:
        img, binary = page(seed, H, W, flat)
        o_img, o_bin, o_orig, stage1 = lib_prepare(img, binary, tgt, lh, mw)
        out[name + "/image"] = img
        out[name + "/binary"] = binary
        out[name + "/params"] = np.array([tgt, lh, -1 if mw is None else mw], np.int64)
        out[name + "/out_img"] = o_img
        out[name + "/out_bin"] = o_bin
        out[name + "/out_orig_bin"] = o_orig
        out[name + "/stage1"] = stage1
    # two-valued image: anti-aliasing off
    img, binary = page(6, 48, 64)
    o_img, o_bin, o_orig, stage1 = lib_prepare(binary, binary // 255, 6, 14)
    out["twoval/image"], out["twoval/binary"] = binary, binary // 255
    out["twoval/params"] = np.array([6, 14, -1], np.int64)
    out["twoval/out_img"], out["twoval/out_bin"], out["twoval/out_orig_bin"], out["twoval/stage1"] = o_img, o_bin, o_orig, stage1
    # preserving_resize (lib/util.py:21-29) on a label map, up and down
    rng = np.random.default_rng(7)
    lab = rng.integers(0, 6, (37, 53)).astype(np.int64)
    for nm, shp in (("pr_up", (91, 120)), ("pr_down", (13, 22))):
        out[nm + "/in"] = lab
        out[nm + "/out"] = resize(lab, shp, order=0, anti_aliasing=False, preserve_range=True)
    # the Gaussian kernels scipy builds for a few sigmas (bit patterns matter, see oracle/resize.py)
    for i, s in enumerate((0.25, 0.9166666666666667, 1.4166666666666665, 2.5)):
        r = int(4.0 * s + 0.5)
        out["gauss/%d" % i] = np.concatenate([[s, r], _gaussian_kernel1d(s, 0, r)[::-1]])
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "resize_vectors.npz")
    np.savez_compressed(path, **{k.replace("/", "__"): v for k, v in out.items()})
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
