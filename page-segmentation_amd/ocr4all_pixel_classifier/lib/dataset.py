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


def _nearest_rescale(a, scale):
    """scale_binary (lib/dataset.py:114-119): skimage rescale(order=0): output shape =
    round(shape * scale), nearest gather."""
    from .util import preserving_resize
    H, W = a.shape[:2]
    out = (int(np.round(H * scale)), int(np.round(W * scale)))
    return preserving_resize(a, out)


def scale_binary(binary, scale):
    return _nearest_rescale(np.asarray(binary), scale)


def scale_image(img, target_shape):
    """lib/dataset.py:122-128: bicubic resize with Gaussian anti-aliasing when the image has more
    than two distinct values.  The bicubic / Gaussian GPU kernels are a later SURVEY 8(f) row;
    until they land this raises instead of silently computing on the CPU."""
    img = np.asarray(img)
    if tuple(img.shape[:2]) == tuple(target_shape):
        return img.astype(np.float64)
    raise Exception("line-height rescaling (bicubic + Gaussian anti-aliasing) is not built yet in the "
                    "MI355X engine: pass pages already normalised to the target line height "
                    "(line_height_px == target_line_height)")


def prepare_images(image, binary, target_line_height, line_height_px, max_width=None, keep_orig_bin=False):
    """lib/dataset.py:131-150.  image: gray uint8 scan (ink dark); binary: 0/255 or 0/1 with
    paper = 1 (ink = 0).  Returns the inverted network input (ink bright), ink = 1 binary."""
    scale = target_line_height / line_height_px
    binary = np.asarray(binary)
    orig_bin = binary / 255 if np.max(binary) > 1 else binary
    bin_ = 1.0 - scale_binary(orig_bin, scale)
    img = 1.0 - scale_image(image, bin_.shape) / 255
    if max_width is not None:
        n_scale = max_width / bin_.shape[1]
        if n_scale < 1.0:
            bin_ = scale_binary(bin_, n_scale)
            img = scale_image(img, bin_.shape)
    img = (img * 255).astype(np.uint8)
    bin_ = bin_.astype(np.uint8)
    if keep_orig_bin:
        return img, bin_, (1 - orig_bin).astype(np.uint8)
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
