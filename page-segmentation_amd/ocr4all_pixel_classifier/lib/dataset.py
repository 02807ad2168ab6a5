"""Dataset records and the dataset-JSON loader (reference: lib/dataset.py).

SingleData / Dataset keep the reference's field names and order (lib/dataset.py:17-41) because
callers build them positionally and from the dataset JSON ({"train": [...], "test": [...],
"eval": [...]} of {binary_path, image_path, mask_path, line_height_px}, README.md:46-70).
Records stay picklable: no GPU handles live in them.
"""
import json
from dataclasses import dataclass
from typing import Any, List, Optional, Tuple

import numpy as np

from .colors import ColorMap


@dataclass
class SingleData:
    image: np.ndarray = None
    binary: Optional[np.ndarray] = None
    orig_binary: Optional[np.ndarray] = None
    mask: np.ndarray = None
    image_path: Optional[str] = None
    binary_path: Optional[str] = None
    mask_path: Optional[str] = None
    line_height_px: Optional[int] = 1
    original_shape: Tuple[int, int] = None
    output_path: Optional[str] = None
    user_data: Any = None


@dataclass
class Dataset:
    data: List[SingleData]
    color_map: ColorMap

    def __len__(self):
        return len(self.data)

    def __iter__(self):
        return iter(self.data)


def scale_binary(binary, scale):
    """lib/dataset.py:114-119: skimage rescale(order=0, no anti-aliasing): output shape =
    np.round(shape * scale), nearest gather on the GPU; float64 result."""
    from pseg_amd import engine as _eng
    from .util import preserving_resize
    binary = np.asarray(binary)
    return preserving_resize(binary, _eng.rescale_shape(binary.shape, scale))


def scale_image(img, target_shape):
    """lib/dataset.py:122-128: bicubic resize, Gaussian anti-aliasing when the image has more
    than two distinct values (GPU: pseg_scale_image); float64 result."""
    from pseg_amd import engine as _eng
    return _eng.scale_image(np.asarray(img), target_shape)


def prepare_images(image, binary, target_line_height, line_height_px, max_width=None, keep_orig_bin=False):
    """lib/dataset.py:131-150.  image: gray uint8 scan (ink dark); binary: 0/255 or 0/1 with
    paper = 1 (ink = 0).  Returns the inverted network input (ink bright), ink = 1 binary.  One
    GPU call (pseg_prepare_images) does the whole chain: nearest rescale of the binary, Gaussian
    anti-aliasing + bicubic resize of the image, optional max_width stage, inversion, uint8."""
    from pseg_amd import engine as _eng
    scale = target_line_height / line_height_px
    img, bin_, orig = _eng.prepare_images(image, binary, scale, max_width)
    if keep_orig_bin:
        return img, bin_, orig
    return img, bin_


def _imread_gray(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert("L"))


def _imread_bin(path):
    """ocr4all.files.imread_bin(path, True) stand-in: gray read, > 127 -> 255 else 0."""
    g = _imread_gray(path)
    return np.where(g > 127, 255, 0).astype(np.uint8)


class DatasetLoader:
    def __init__(self, target_line_height, color_map: ColorMap, prediction=False, max_width=None):
        self.target_line_height = target_line_height
        self.prediction = prediction
        self.color_map = color_map
        self.max_width = max_width

    def load_images(self, entry: SingleData) -> SingleData:
        """lib/dataset.py:160-191.  The reference derives the binary from the *image* attribute /
        path (attr 'image', :172) and never reads binary_path; kept."""
        img = entry.image if entry.image is not None else _imread_gray(entry.image_path)
        original_shape = img.shape
        bin_ = entry.image if entry.image is not None else _imread_bin(entry.image_path)
        img, bin_, orig_bin = prepare_images(img, bin_, self.target_line_height, entry.line_height_px,
                                             self.max_width, keep_orig_bin=True)
        if not self.prediction:
            from .util import preserving_resize
            mask = entry.mask if entry.mask is not None else self.color_map.imread_labels(entry.mask_path)
            mask = preserving_resize(mask, img.shape)
            assert mask.shape == img.shape
            entry.mask = mask.astype(np.uint8)
        entry.binary = bin_
        entry.orig_binary = orig_bin
        entry.image = img
        entry.original_shape = original_shape
        return entry

    def load_data(self, all_dataset_files) -> Dataset:
        return Dataset([self.load_images(d) for d in all_dataset_files], self.color_map)

    def load_data_from_json(self, files, type) -> Dataset:
        """lib/dataset.py:200-208."""
        entries = []
        for f in files:
            with open(f, 'r') as fh:
                js = json.load(fh)
            kinds = ["train", "test", "eval"] if type == "all" else [type]
            for t in kinds:
                entries += [SingleData(**d) for d in js[t]]
        print(f"Loading {len(entries)} data of type {type}")
        return self.load_data(entries)
