"""Prediction / PredictSettings -- the records Predictor takes and yields (reference API: lib/predictor_data.py:12-26;
same names, order and defaults, which is all the frontend relies on).  Built from field tables."""
import collections
import dataclasses
from typing import Callable, List, Optional

import numpy as np

from .colors import ColorMap
from .dataset import SingleData

#: one predicted page: label map (H,W) int64, probabilities (H,W,C) float32, the input record
Prediction = collections.namedtuple("Prediction", ("labels", "probabilities", "data"))
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
