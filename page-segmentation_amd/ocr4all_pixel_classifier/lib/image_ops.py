"""compute_char_height (reference: lib/image_ops.py:58-82) on the GPU: Otsu histogram,
binarise, union-find CCL, bounding boxes; the glyph filter and the upper median run on the host
over the component list."""
import os

import numpy as np

from pseg_amd import engine


def compute_char_height(file_name: str, inverse: bool):
    if not os.path.exists(file_name):
        raise Exception(f"File does not exist at {file_name}")
    from PIL import Image
    gray = np.asarray(Image.open(file_name).convert("L"))
    height, _ = engine.otsu_char_height(gray, inverse)
    return height
