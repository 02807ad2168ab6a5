"""CPU restatement of the line-height normalisation path (TEST INFRASTRUCTURE ONLY -- the product
never imports this): lib/dataset.py:114-150 (scale_binary, scale_image, prepare_images) and
lib/util.py:21-29 (preserving_resize).

The arithmetic lives in third-party code absent from /root/reference: scikit-image 0.17.2
(requirements.txt:131) `transform.resize/rescale` -> `scipy.ndimage.gaussian_filter` (anti-aliasing)
-> `warp` -> `_warp_fast` (nearest / Catmull-Rom bicubic, 'reflect' boundary, clip to the input
range).  Restated here from the published algorithm:

* output shape of rescale = np.round(scale * shape) (half to even);
* anti-aliasing (only when the image has > 2 distinct values): sigma = max(0, (in/out - 1) / 2) per
  axis, kernel radius int(4 sigma + 0.5), weights exp(-x^2 / 2 sigma^2) / sum; axis 0 then axis 1;
  boundary 'mirror' (no edge repeat); accumulation order of scipy's symmetric correlate1d: centre
  tap first, then the tap pairs from the outermost inwards, (x[-j] + x[+j]) * w[j]; the filter keeps
  the dtype of its input: for a uint8 image every pass truncates to uint8;
* warp: input coordinate = f * o + (f / 2 - 0.5) with f = in / out per axis (skimage estimates this
  affine by least squares, so its coefficients carry ~1e-16 platform-dependent noise; the exact ones
  are used here); order 0: round half away from zero; order 3: 4x4 Catmull-Rom around floor(coord),
  columns first then rows, 'reflect' index mapping; float64 throughout; result clipped to the
  [min, max] of the (filtered) input.

PARITY: pinned against scikit-image 0.18.3 / scipy 1.7.1 run in the build container
(tests/golden/make_resize_golden.py -> tests/golden/resize_vectors.npz); the reference's pinned
0.17.2 itself is not installable offline, and the reference holds no fixture for this path."""
import numpy as np


def rescale_shape(shape, scale):
    return tuple(int(v) for v in np.round(scale * np.asarray(shape[:2])))


def gaussian_kernel(sigma):
    radius = int(4.0 * sigma + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    return (phi / phi.sum())[::-1].copy(), radius


def aa_sigmas(in_shape, out_shape):
    f = np.asarray(in_shape[:2], dtype=float) / np.asarray(out_shape[:2], dtype=float)
    return np.maximum(0, (f - 1) / 2)


def _mirror_index(i, n):
    if n == 1:
        return np.zeros_like(i)
    p = 2 * (n - 1)
    i = np.mod(i, p)
    return np.where(i >= n, p - i, i)


def correlate1d_mirror(a, w, radius, axis):
    """a: float64 2-D; returns float64 (before any cast)."""
    a = np.moveaxis(a, axis, 0)
    n = a.shape[0]
    idx = np.arange(n)
    acc = a * w[radius]
    for j in range(radius, 0, -1):
        lo = a[_mirror_index(idx - j, n)]
        hi = a[_mirror_index(idx + j, n)]
        acc = acc + (lo + hi) * w[radius - j]
    return np.moveaxis(acc, 0, axis)


def gaussian_filter(img, sigmas):
    """scipy.ndimage.gaussian_filter(img, sigmas, mode='mirror'): output dtype = input dtype."""
    out = img
    for axis in (0, 1):
        s = float(sigmas[axis])
        if s <= 1e-15:
            continue
        w, r = gaussian_kernel(s)
        res = correlate1d_mirror(out.astype(np.float64), w, r, axis)
        out = res.astype(img.dtype) if img.dtype != np.float64 else res     # C cast: truncation
    return out


def _warp_coords(n_in, n_out):
    f = float(n_in) / float(n_out)
    return f * np.arange(n_out, dtype=np.float64) + (f * 0.5 - 0.5)


def resize_nearest(a, out_shape):
    """order 0, no anti-aliasing, preserve_range: float64 result."""
    a = np.asarray(a)
    H, W = a.shape[:2]
    Ho, Wo = int(out_shape[0]), int(out_shape[1])

    def rnd(c):
        return np.where(c > 0, c + 0.5, c - 0.5).astype(np.int64)       # truncation toward zero
    r = _mirror_index(rnd(_warp_coords(H, Ho)), H)
    c = _mirror_index(rnd(_warp_coords(W, Wo)), W)
    return a[r][:, c].astype(np.float64)


def _cubic(x, f0, f1, f2, f3):
    return f1 + 0.5 * x * (f2 - f0 + x * (2.0 * f0 - 5.0 * f1 + 4.0 * f2 - f3 + x * (3.0 * (f1 - f2) + f3 - f0)))


def resize_bicubic(a, out_shape):
    """order 3 warp of a float64 image, 'reflect', clipped to the input range."""
    a = np.asarray(a, np.float64)
    H, W = a.shape
    Ho, Wo = int(out_shape[0]), int(out_shape[1])
    yr, xc = _warp_coords(H, Ho), _warp_coords(W, Wo)
    r0 = np.floor(yr).astype(np.int64)
    c0 = np.floor(xc).astype(np.int64)
    tr = (yr - r0)[:, None]
    tc = (xc - c0)[None, :]
    rows = [_mirror_index(r0 - 1 + k, H) for k in range(4)]
    cols = [_mirror_index(c0 - 1 + k, W) for k in range(4)]
    fr = []
    for k in range(4):
        g = a[rows[k]]
        fr.append(_cubic(tc, g[:, cols[0]], g[:, cols[1]], g[:, cols[2]], g[:, cols[3]]))
    out = _cubic(tr, fr[0], fr[1], fr[2], fr[3])
    return np.clip(out, a.min(), a.max())


def scale_binary(binary, scale):                      # lib/dataset.py:114-119
    return resize_nearest(binary, rescale_shape(np.asarray(binary).shape, scale))


def scale_image(img, target_shape):                   # lib/dataset.py:122-128
    img = np.asarray(img)
    if len(np.unique(img)) > 2:
        img = gaussian_filter(img, aa_sigmas(img.shape, target_shape))
    return resize_bicubic(img.astype(np.float64), target_shape)


def prepare_images(image, binary, target_line_height, line_height_px, max_width=None):
    """lib/dataset.py:131-150 -> (img uint8, bin uint8, orig_bin uint8, stage-1 bicubic float64)."""
    scale = target_line_height / line_height_px
    orig_bin = binary / 255 if np.max(binary) > 1 else binary
    b = 1.0 - scale_binary(orig_bin, scale)
    stage1 = scale_image(image, b.shape)
    img = 1.0 - stage1 / 255
    if max_width is not None:
        n_scale = max_width / b.shape[1]
        if n_scale < 1.0:
            b = scale_binary(b, n_scale)
            img = scale_image(img, b.shape)
    return (img * 255).astype(np.uint8), b.astype(np.uint8), (1 - orig_bin).astype(np.uint8), stage1
