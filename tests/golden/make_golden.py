#!/usr/bin/env python3
"""Generates tests/golden/reference_vectors.json from the REFERENCE ITSELF, in the dev container.

Only two modules of /root/reference import here (everything else needs TensorFlow / OpenCV /
scikit-image / ocr4all-pylib, absent and not installable offline -- SURVEY.md section 8c):
  ocr4all_pixel_classifier/lib/util.py          (gray_to_rgb, image_to_batch)
  ocr4all_pixel_classifier/lib/architecture.py  (default_preprocess, Architecture / Optimizers values)
plus the reference's one runnable third-party call on the predict path, scipy.special.softmax
(lib/network.py:249,258).  The vectors are data (inputs + expected outputs); nothing of the
reference's source travels.  Run:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

REF = "/root/reference"
sys.path.insert(0, REF)
from ocr4all_pixel_classifier.lib import architecture, util  # noqa: E402
from scipy.special import softmax  # noqa: E402

out = {}
# default_preprocess on every byte value, then the float32 cast Keras applies at predict_on_batch
u = np.arange(256, dtype=np.uint8)
pre = architecture.default_preprocess(u)
assert pre.dtype == np.float64
out["preprocess_f32_bits"] = pre.astype(np.float32).view(np.uint32).tolist()
out["architecture_values"] = {m.name: m.value for m in architecture.Architecture}
out["optimizer_values"] = {m.name: m.value for m in architecture.Optimizers}
# util.image_to_batch / gray_to_rgb shapes and values
rng = np.random.default_rng(0)
img = rng.integers(0, 256, size=(3, 4), dtype=np.uint8)
out["util_img"] = img.tolist()
out["image_to_batch_shape_2d"] = list(util.image_to_batch(img).shape)
out["image_to_batch_shape_3d"] = list(util.image_to_batch(np.zeros((3, 4, 3))).shape)
out["gray_to_rgb"] = util.gray_to_rgb(img).tolist()
out["gray_to_rgb_passthrough_shape"] = list(util.gray_to_rgb(np.zeros((3, 4, 3), np.uint8)).shape)
# scipy softmax on float32 logits (the reference's call) for a few rows incl. ties / large values
z = np.array([[0.1, -0.2, 0.3], [5.0, 5.0, -1.0], [30.0, -30.0, 0.0], [0.0, 0.0, 0.0],
              [-1e-3, 1e-3, 0.0], [12.5, 11.25, 13.0]], np.float32)
out["softmax_logits_bits"] = z.view(np.uint32).tolist()
out["softmax_probs_bits"] = softmax(z, -1).astype(np.float32).view(np.uint32).tolist()
out["argmax"] = np.argmax(z, -1).tolist()

path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_vectors.json")
with open(path, "w") as f:
    json.dump(out, f, indent=0, sort_keys=True)
print("wrote", path)
