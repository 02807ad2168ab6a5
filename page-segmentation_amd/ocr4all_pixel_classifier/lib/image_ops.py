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


def _fg_counts(pred, mask, bin, n_classes):
    """counts[b][m][p] from the GPU (pseg_eval_confusion); labels outside [0, n_classes) land in the last slot."""
    return engine.eval_confusion(pred, mask, bin, n_classes)


def fgpa(pred: np.ndarray, mask: np.ndarray, bin: np.ndarray):
    """Foreground pixel accuracy (lib/image_ops.py:8-19): pred*bin != mask*bin happens exactly on the ink pixels
    whose labels differ, so the joint histogram's ink plane gives both counts."""
    pred, mask = np.asarray(pred), np.asarray(mask)
    top = int(max(pred.max(initial=0), mask.max(initial=0))) + 1
    if pred.min(initial=0) < 0 or mask.min(initial=0) < 0 or top > 255:
        raise Exception("fgpa: labels must lie in 0..254")
    c = _fg_counts(pred, mask, bin, top)[1]
    fg_count = int(c.sum())
    wrong = fg_count - int(np.trace(c))
    return (fg_count - wrong) / fg_count                      # ZeroDivisionError on a page without ink, as the reference


def fgoverlap_per_class(pred: np.ndarray, mask: np.ndarray, bin: np.ndarray, n_classes: int):
    """Per-class foreground overlap (lib/image_ops.py:22-55): index i for class i, i = 0..n_classes (the
    reference's range(n_classes + 1)); (overlaps, tps, fps, fns).  bin must be 0/1 as the reference documents:
    (label + 1) * bin - 1 is the label on ink and -1 (no class) on paper."""
    bin = np.asarray(bin)
    if bin.size and bin.max() > 1:
        raise Exception("fgoverlap_per_class: bin must be 0/1 (1 is foreground)")
    k = int(n_classes) + 1                                    # classes 0..n_classes are reported
    c_full = _fg_counts(pred, mask, bin, k)[1]                 # ink plane, [mask][pred]; slot k = other labels
    overlaps, tps, fps, fns = [], [], [], []
    for i in range(k):
        tp = int(c_full[i, i])
        fp = int(c_full[:, i].sum()) - tp                      # predicted i, expected something else
        fn = int(c_full[i, :].sum()) - tp
        if tp + fp + fn == 0:
            overlaps.append(np.nan); tps.append(0); fps.append(0); fns.append(0)
        else:
            overlaps.append(tp / (tp + fp + fn)); tps.append(tp); fps.append(fp); fns.append(fn)
    return overlaps, tps, fps, fns
