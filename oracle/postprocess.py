"""Oracle for the integer pre/post-process (TEST INFRASTRUCTURE ONLY).

The reference does this with OpenCV / scikit-image / ocr4all-pylib, none of which is installed
here; connected components come from scipy.ndimage.label (independent implementation, 4-conn
default structure) and everything else is restated in NumPy.  PARITY UNPINNED versus cv2 itself;
the algorithms are label-order independent, so any correct 4-connected labelling gives the same
result.
"""
import numpy as np
from scipy import ndimage


def vote_connected_component_class(pred, binary):
    """lib/postprocess.py:9-26.  4-connected components of `binary` (non-zero = ink); every
    component's pixels take the component's most frequent predicted class; ties -> lowest class
    (np.argmax over the bincount).  Returns a new int64 array (the reference mutates in place)."""
    pred = np.asarray(pred).astype(np.int64)
    labels, n = ndimage.label(np.asarray(binary) != 0)
    if n == 0:
        return pred.copy()
    ncls = int(pred.max()) + 1
    fg = labels > 0
    hist = np.bincount((labels[fg] - 1) * ncls + pred[fg], minlength=n * ncls).reshape(n, ncls)
    winner = np.argmax(hist, axis=1)
    out = pred.copy()
    out[fg] = winner[labels[fg] - 1]
    return out


def add_bounding_boxes(pred):
    """lib/postprocess.py:29-42.  For every class c in ascending order, every 4-connected
    component of (pred == c) paints its bounding box with c into a zero image; later classes
    overwrite earlier ones."""
    pred = np.asarray(pred).astype(np.int64)
    new = np.zeros_like(pred)
    for c in np.unique(pred):
        labels, n = ndimage.label(pred == c)
        for sl in ndimage.find_objects(labels):
            if sl is not None:
                new[sl] = c
    return new


def generate_output_masks(pred, binary, lut):
    """lib/output.py:44-60.  lut: (n_labels,3) uint8 label->RGB table (ColorMap.to_rgb_array).
    Returns (color, overlay, inverted_overlay, fg_color_mask), each (H,W,3) uint8."""
    pred = np.asarray(pred)
    binary = np.asarray(binary)
    color = np.asarray(lut, np.uint8)[pred]
    # uint8 arithmetic exactly as the reference: (1 - binary) wraps for binary > 1
    foreground = (1 - binary)
    overlay = color.copy()
    overlay[foreground == 0] = 0
    inverted = color.copy()
    inverted[binary == 0] = 0
    fg = color.copy()
    fg[foreground != 0] = 0
    return color, overlay, inverted, fg


def otsu_threshold(gray_u8):
    """cv2.threshold(..., THRESH_OTSU) threshold value, restated from OpenCV 4.5.5
    modules/imgproc/src/thresh.cpp getThreshVal_Otsu_8u (published algorithm): maximise
    q1*q2*(mu1-mu2)^2 over t in 0..255, first maximum wins, double precision."""
    h = np.bincount(np.asarray(gray_u8, np.uint8).ravel(), minlength=256).astype(np.float64)
    n = h.sum()
    scale = 1.0 / n
    mu = float((np.arange(256) * h).sum() * scale)
    mu1 = 0.0
    q1 = 0.0
    max_sigma = 0.0
    max_val = 0
    eps = np.finfo(np.float32).eps
    for i in range(256):
        p_i = h[i] * scale
        mu1 *= q1
        q1 += p_i
        q2 = 1.0 - q1
        if min(q1, q2) < eps or max(q1, q2) > 1.0 - eps:
            continue
        mu1 = (mu1 + i * p_i) / q1
        mu2 = (mu - q1 * mu1) / q2
        sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2)
        if sigma > max_sigma:
            max_sigma = sigma
            max_val = i
    return max_val


def compute_char_height_from_gray(gray_u8, inverse=False):
    """lib/image_ops.py:58-82 minus the file read: Otsu binarise (pixel > t -> 255), invert
    unless `inverse`, 4-connected components, keep 0.5 < w/h < 2, 10 < h < 60, 5 < w < 50, return
    the UPPER median height (sorted[int(len/2)]) or None."""
    g = np.asarray(gray_u8, np.uint8)
    t = otsu_threshold(g)
    img = np.where(g > t, 255, 0).astype(np.uint8)
    if not inverse:
        img = 255 - img
    labels, n = ndimage.label(img != 0)
    hs = []
    for sl in ndimage.find_objects(labels):
        h = sl[0].stop - sl[0].start
        w = sl[1].stop - sl[1].start
        if 0.5 < (w / h) < 2 and 10 < h < 60 and 5 < w < 50:
            hs.append(h)
    hs.sort()
    if not hs:
        return None
    return hs[int(len(hs) / 2)]


def nearest_resize(a, out_shape):
    """skimage.transform.resize(order=0, anti_aliasing=False, preserve_range=True) as used by
    lib/util.py:21-29 and lib/dataset.py:182: output pixel (r,c) samples input at
    ((r+0.5)*H/Ho-0.5, (c+0.5)*W/Wo-0.5), rounded half-away-from... skimage's order-0 warp
    rounds the coordinate to nearest (round-half-up on the +0.5 form) and clamps (mode='edge'
    behaviour at the border for order 0).  PARITY UNPINNED vs scikit-image 0.17.2 (absent)."""
    a = np.asarray(a)
    H, W = a.shape[:2]
    Ho, Wo = out_shape
    r = np.floor((np.arange(Ho) + 0.5) * (H / Ho)).astype(np.int64).clip(0, H - 1)
    c = np.floor((np.arange(Wo) + 0.5) * (W / Wo)).astype(np.int64).clip(0, W - 1)
    return a[r][:, c]
