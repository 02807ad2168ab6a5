"""Synthetic pages, masks and weights -- no datasets or checkpoints exist offline.

Not arithmetic under test: this only manufactures inputs.  Page i is a pure function of
numpy.random.default_rng(1000 + i); weights of default_rng(seed) in Keras creation order.
"""
from collections import OrderedDict

import numpy as np


def synth_page(page_index, H=2048, W=1536, n_classes=3):
    """-> (image uint8 (H,W) inverted gray [ink bright], binary uint8 {0,1} [ink=1],
           mask uint8 (H,W) class ids)."""
    rng = np.random.default_rng(1000 + int(page_index))
    page = np.clip(rng.normal(225.0, 8.0, size=(H, W)), 0, 255).astype(np.uint8)
    mask = np.zeros((H, W), np.uint8)
    margin = min(96, max(4, W // 8), max(4, H // 8))
    # two "image" rectangles of mid-gray texture
    rects = []
    for _ in range(2):
        rh = int(rng.integers(max(8, H // 10), max(9, H // 4)))
        rw = int(rng.integers(max(8, W // 8), max(9, W // 3)))
        y0 = int(rng.integers(margin, max(margin + 1, H - margin - rh)))
        x0 = int(rng.integers(margin, max(margin + 1, W - margin - rw)))
        y1, x1 = min(H, y0 + rh), min(W, x0 + rw)
        rects.append((y0, x0, y1, x1))
    # text lines every 48 px
    line_id = 0
    for ly in range(margin, H - margin - 30, 48):
        x = margin
        x_end = W - margin
        kind = 1
        if n_classes >= 6:
            if line_id == 0:
                kind = 3                       # heading
            elif ly + 48 >= H - margin - 30:
                kind = 5                       # page number line
        first_x = None
        while x < x_end - 20:
            gw = int(rng.integers(6, 21))
            gh = int(rng.integers(10, 31))
            gap = int(rng.integers(2, 7))
            inside = any(y0 - 30 <= ly <= y1 and x0 - 20 <= x <= x1 for (y0, x0, y1, x1) in rects)
            if not inside:
                blob = np.clip(rng.normal(40.0, 15.0, size=(gh, gw)), 0, 255).astype(np.uint8)
                yy = ly + (30 - gh)
                page[yy:yy + gh, x:x + gw] = blob
                if first_x is None:
                    first_x = x
                k = kind
                if n_classes >= 6 and x < margin + W // 12 and kind == 1 and line_id % 7 == 3:
                    k = 4                      # marginalia
                mask[ly:ly + 30, x:x + gw + gap] = k
            x += gw + gap
        line_id += 1
    for (y0, x0, y1, x1) in rects:
        tex = np.clip(rng.normal(120.0, 30.0, size=(y1 - y0, x1 - x0)), 0, 255).astype(np.uint8)
        page[y0:y1, x0:x1] = tex
        mask[y0:y1, x0:x1] = 2
    binary = (page < 128).astype(np.uint8)
    image = (255 - page).astype(np.uint8)      # lib/dataset.py:137: ink bright
    return image, binary, mask


def _namer():
    counts = {}

    def nm(base):
        i = counts.get(base, 0)
        counts[base] = i + 1
        return base if i == 0 else "%s_%d" % (base, i)
    return nm


def glorot_weights(specs, seed=42, gain=1.0, bias_scale=0.0):
    """specs: [(name, shape)] as returned by Engine.weight_specs() (Keras creation order).
    Keras glorot_uniform: limit = sqrt(6 / (fan_in + fan_out)), fans = receptive field x channels."""
    rng = np.random.default_rng(seed)
    out = OrderedDict()
    for name, shp in specs:
        if name.endswith("/kernel"):
            rf = shp[0] * shp[1]
            limit = np.sqrt(6.0 / (rf * shp[2] + rf * shp[3])) * gain
            out[name] = rng.uniform(-limit, limit, size=shp).astype(np.float32)
        elif name.endswith(("/gamma", "/moving_variance")):
            # BatchNormalization initialisers are ones / zeros; bias_scale > 0 perturbs them so that every term counts
            out[name] = (1.0 + (rng.uniform(-0.3, 0.3, size=shp) if bias_scale > 0 else 0.0) * np.ones(shp)).astype(np.float32)
        else:
            if bias_scale > 0:
                out[name] = rng.uniform(-bias_scale, bias_scale, size=shp).astype(np.float32)
            else:
                out[name] = np.zeros(shp, np.float32)
    return out
