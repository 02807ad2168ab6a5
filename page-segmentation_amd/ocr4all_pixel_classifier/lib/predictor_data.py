"""Prediction / PredictSettings -- the records Predictor takes and yields (reference API: lib/predictor_data.py:12-26;
same names, order and defaults, which is all the frontend relies on).  Built from field tables."""
import collections
import dataclasses
from typing import Callable, List, Optional

import numpy as np

from .colors import ColorMap
from .dataset import SingleData

_PredictionFields = collections.namedtuple("Prediction", ("labels", "probabilities", "data"))


class LazyArray:
    """A probabilities map that is computed when first read: 12 bytes per pixel and class leave the device only for
    callers that look at them (the label map is what Predictor's consumers use)."""

    def __init__(self, thunk):
        self._thunk, self._value = thunk, None

    def get(self):
        if self._thunk is not None:
            self._value, self._thunk = self._thunk(), None
        return self._value


class Prediction(_PredictionFields):
    """One predicted page: label map (H,W) int64, probabilities (H,W,C) float32, the input record
    (lib/predictor_data.py:12-15: a NamedTuple of these three).  `labels` (the device chain hands down the compact uint8
    map; the int64 form of the reference is made on first read) and `probabilities` may be handed in as LazyArrays; they
    are resolved on attribute access, indexing and unpacking, so readers always see an ndarray."""
    __slots__ = ()

    @property
    def labels(self):
        v = tuple.__getitem__(self, 0)
        return v.get() if isinstance(v, LazyArray) else v

    @property
    def probabilities(self):
        v = tuple.__getitem__(self, 1)
        return v.get() if isinstance(v, LazyArray) else v

    def __getitem__(self, i):
        v = tuple.__getitem__(self, i)
        if isinstance(i, slice):
            return tuple(x.get() if isinstance(x, LazyArray) else x for x in v)
        return v.get() if isinstance(v, LazyArray) else v

    def __iter__(self):
        for v in tuple.__iter__(self):
            yield v.get() if isinstance(v, LazyArray) else v


Prediction.__annotations__ = {"labels": np.ndarray, "probabilities": np.ndarray, "data": SingleData}

_PostProcessors = Optional[List[Callable[[np.ndarray, SingleData], np.ndarray]]]

PredictSettings = dataclasses.make_dataclass("PredictSettings", [
    # (field, type, default)        network: model path; output: directory for masks; color_map: only for coloured images
    ("network", str, dataclasses.field(default=None)),
    ("output", str, dataclasses.field(default=None)),
    ("high_res_output", bool, dataclasses.field(default=False)),
    ("color_map", Optional[ColorMap], dataclasses.field(default=None)),
    ("n_classes", int, dataclasses.field(default=-1)),
    ("post_process", _PostProcessors, dataclasses.field(default=None)),
    ("gpu_allow_growth", bool, dataclasses.field(default=False)),
])
PredictSettings.__module__ = __name__
