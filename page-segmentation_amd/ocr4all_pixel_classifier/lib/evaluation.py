"""Evaluation helpers (reference: lib/evaluation.py).  The reductions over whole label maps and the connected
component tables run on the GPU (pseg_eval_confusion, pseg_cc_label, pseg_cc_tables); the per-component
callbacks of ConnectedComponentEval.run_per_component receive the same arrays, in the same order, as the
reference's `bbox(image)[component]` slices, cut out of one GPU-sorted pixel list."""
from typing import Callable, Generator, Tuple, TypeVar, Union

import numpy as np

from pseg_amd import engine as _eng
from .cc import cc_bbox_func


def _counts(mask, pred):
    mask, pred = np.asarray(mask), np.asarray(pred)
    top = int(max(mask.max(initial=0), pred.max(initial=0))) + 1
    if mask.min(initial=0) < 0 or pred.min(initial=0) < 0 or top > 255:
        raise Exception("labels must lie in 0..254")
    return _eng.eval_confusion(pred, mask, None, top)[1]             # [mask][pred]


def count_matches(mask: np.ndarray, pred: np.ndarray, label: int) -> Tuple[int, int, int]:
    """lib/evaluation.py:8-22, names as there: (mask & pred, mask & ~pred, ~mask & pred) for `label`."""
    c = _counts(mask, pred)
    if not 0 <= label < c.shape[0]:
        return 0, 0, 0
    tp = int(c[label, label])
    return tp, int(c[label].sum()) - tp, int(c[:, label].sum()) - tp


def total_accuracy(mask: np.ndarray, pred: np.ndarray) -> Tuple[int, int]:
    c = _counts(mask, pred)
    return int(np.trace(c)), int(np.asarray(mask).size)


def f1_measures(tp: int, fp: int, fn: int) -> Tuple[float, float, float]:
    if tp == 0:
        return 0.0, 0.0, 0.0
    precision = tp / (tp + fp)
    recall = tp / (tp + fn)
    return precision, recall, f1(precision, recall)


def f1(precision: float, recall: float) -> float:
    return 2 * precision * recall / (precision + recall)


class _CcEqual:
    """cc_equal(threshold): callable on (pred, mask) slices like the reference's lambda; carries its parameters so
    run_per_component can answer it from the GPU tables without touching the pixels."""

    def __init__(self, threshold):
        self.threshold = threshold

    def __call__(self, pred, mask):
        return np.count_nonzero(pred == mask) / np.size(mask) >= self.threshold

    def from_tables(self, t, i):
        return t["eq"][i] / t["stats"][i, 4] >= self.threshold


def cc_equal(threshold: float):
    return _CcEqual(threshold)


class _CcMatching:
    def __init__(self, label, threshold_tp, threshold_fp, threshold_mask):
        self.label, self.threshold_tp, self.threshold_fp = label, threshold_tp, threshold_fp
        self.threshold_mask = threshold_mask if threshold_mask else threshold_tp

    def _decide(self, n_pred, n_mask, size):
        pred_match_fp = n_pred / size >= self.threshold_fp
        pred_match_tp = n_pred / size >= self.threshold_tp
        mask_match = n_mask / size >= self.threshold_mask
        return np.array([int(pred_match_tp and mask_match), int(pred_match_fp and not mask_match),
                         int(mask_match and not pred_match_tp)])

    def __call__(self, mask, pred):
        return self._decide(np.count_nonzero(pred == self.label), np.count_nonzero(mask == self.label), np.size(mask))

    def from_tables(self, t, i):
        k = t["hist_pred"].shape[1] - 1
        n_pred = int(t["hist_pred"][i, self.label]) if 0 <= self.label < k else 0
        n_mask = int(t["hist_mask"][i, self.label]) if 0 <= self.label < k else 0
        return self._decide(n_pred, n_mask, int(t["stats"][i, 4]))


def cc_matching(label: int, threshold_tp: float, threshold_fp: float, threshold_mask: float = None):
    """(1,0,0) for TP, (0,1,0) for FP, (0,0,1) for FN (lib/evaluation.py:56-70)."""
    return _CcMatching(label, threshold_tp, threshold_fp, threshold_mask)


class ConnectedComponentEval:
    def __init__(self, mask: np.ndarray, prediction: np.ndarray, binary_image: np.ndarray, connectivity=4):
        if binary_image.ndim > 2:
            raise ValueError("Binary image must be 2-dimensional")
        self.mask = mask
        self.pred = prediction
        self.binary_image = binary_image
        self.filtered_label = None
        self.threshold = None
        self.num_labels, self.labels = _eng.cc_label(binary_image.astype("uint8"), connectivity)
        m, p = np.asarray(mask), np.asarray(prediction)
        ints = m.dtype.kind in "iub" and p.dtype.kind in "iub" and m.shape == self.labels.shape == p.shape
        top = int(max(m.max(initial=0), p.max(initial=0))) + 1 if ints else 0
        self._tabled = ints and m.min(initial=0) >= 0 and p.min(initial=0) >= 0 and top <= 255
        t = _eng.cc_tables(self.labels, self.num_labels, p if self._tabled else None, m if self._tabled else None,
                           top, want_stats=True, want_order=True)
        self.stats, self.centroids = t["stats"], t["centroids"]
        self._t = t
        self._offsets = np.concatenate([[0], np.cumsum(self.stats[:, 4].astype(np.int64))])

    def only_label(self, label: int, threshold: float):
        self.filtered_label = label
        self.threshold = threshold
        return self

    def _pixels(self, image, i):
        """bbox(image)[bbox(labels) == i]: the component's pixels in raster order."""
        o = self._t["order"][self._offsets[i]:self._offsets[i + 1]]
        return np.asarray(image).reshape(-1, *np.asarray(image).shape[2:])[o]

    def _label_ratio_i(self, image, i):
        px = self._pixels(image, i)
        return np.count_nonzero(px == self.filtered_label) / np.size(px)

    def _filter(self, component: Union[int, np.ndarray], bbox=None):
        if not self.filtered_label:
            return True
        if not isinstance(component, (int, np.integer)):
            raise Exception("_filter takes the component index on the GPU build")
        i = int(component)
        return self._label_ratio_i(self.mask, i) >= self.threshold or self._label_ratio_i(self.pred, i) > 0

    def _call_masked(self, component: int, func, bbox=None):
        i = int(component)
        return func(self._pixels(self.mask, i), self._pixels(self.pred, i))

    T = TypeVar('T')

    def run_per_component(self, func: Callable[[np.ndarray, np.ndarray], T]) -> Generator[T, None, None]:
        fast = self._tabled and hasattr(func, "from_tables")
        if self.filtered_label and self._tabled:
            k = self._t["hist_mask"].shape[1] - 1
            lab = self.filtered_label
            area = np.maximum(self.stats[:, 4], 1)
            nm = self._t["hist_mask"][:, lab] if 0 <= lab < k else np.zeros(self.num_labels, np.int64)
            np_ = self._t["hist_pred"][:, lab] if 0 <= lab < k else np.zeros(self.num_labels, np.int64)
            keep = (nm / area >= self.threshold) | (np_ / area > 0)
        else:
            keep = None
        for i in range(1, self.num_labels):
            if keep is not None:
                if not keep[i]:
                    continue
            elif not self._filter(i):
                continue
            yield func.from_tables(self._t, i) if fast else self._call_masked(i, func)
