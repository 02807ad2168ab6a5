"""Mask generation and output (reference: lib/output.py)."""
import os
from dataclasses import dataclass, replace
from typing import Optional

import numpy as np

from pseg_amd import engine

from .colors import ColorMap
from .dataset import SingleData


@dataclass
class Masks:
    color: np.ndarray
    overlay: np.ndarray
    inverted_overlay: np.ndarray
    fg_color_mask: Optional[np.ndarray] = None


def generate_output_masks(data: SingleData, pred: np.ndarray, color_map: ColorMap) -> Masks:
    """lib/output.py:44-60 as one streaming GPU kernel (pseg_masks)."""
    color, overlay, inverted, fg = engine.masks(pred, np.asarray(data.binary).astype(np.uint8), color_map.lut())
    return Masks(color=color, overlay=overlay, inverted_overlay=inverted, fg_color_mask=fg)


def output_data(output_dir, pred, data: SingleData, color_map):
    """lib/output.py:20-41: PNG writing stays host-side."""
    from PIL import Image
    if pred.ndim == 3:
        assert pred.shape[0] == 1
        pred = pred[0]
    if data.output_path:
        filename = data.output_path
        d = os.path.dirname(filename)
        if os.path.isabs(d):
            os.makedirs(d, exist_ok=True)
        elif d:
            for category in ("color", "overlay", "inverted"):
                os.makedirs(os.path.join(output_dir, category, d), exist_ok=True)
    else:
        filename = os.path.basename(data.image_path)
    masks = generate_output_masks(data, pred, color_map)
    Image.fromarray(masks.color).save(os.path.join(output_dir, "color", filename))
    Image.fromarray(masks.overlay).save(os.path.join(output_dir, "overlay", filename))
    Image.fromarray(masks.inverted_overlay).save(os.path.join(output_dir, "inverted", filename))


def scale_to_original_shape(data: SingleData, pred):
    """lib/output.py:63-79."""
    from .util import preserving_resize
    resized_image = preserving_resize(data.image, data.original_shape)
    pred = preserving_resize(pred, data.original_shape).astype('int64')
    if data.binary.shape != tuple(data.original_shape):
        if data.orig_binary is not None:
            resized_binary = data.orig_binary
        else:
            resized_binary = preserving_resize(data.binary, data.original_shape).astype('bool')
    else:
        resized_binary = data.binary
    return replace(data, binary=resized_binary, image=resized_image), pred
