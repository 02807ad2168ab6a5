"""Prediction / PredictSettings (reference: lib/predictor_data.py:12-26)."""
from dataclasses import dataclass
from typing import Callable, List, NamedTuple, Optional

import numpy as np

from .colors import ColorMap
from .dataset import SingleData


class Prediction(NamedTuple):
    labels: np.ndarray
    probabilities: np.ndarray
    data: SingleData


@dataclass
class PredictSettings:
    network: str = None
    output: str = None
    high_res_output: bool = False
    color_map: Optional[ColorMap] = None
    n_classes: int = -1
    post_process: Optional[List[Callable[[np.ndarray, SingleData], np.ndarray]]] = None
    gpu_allow_growth: bool = False
