"""Post-processors (reference: lib/postprocess.py), type Callable[[pred, SingleData], pred].
Both run on the GPU through libpseg.so (union-find CCL + vote / bbox kernels)."""
from typing import Callable

import numpy as np

from pseg_amd import engine

from .dataset import SingleData


def vote_connected_component_class(pred: np.ndarray, data: SingleData) -> np.ndarray:
    """lib/postprocess.py:9-26; like the reference, `pred` is modified in place when it is a
    C-contiguous int64 array (np.argmax's output) and returned."""
    if pred.dtype != np.int64 or not pred.flags.c_contiguous:
        pred = np.ascontiguousarray(pred, dtype=np.int64)
    return engine.cc_vote(pred, data.binary)


def add_bounding_boxes(pred: np.ndarray, data: SingleData) -> np.ndarray:
    """lib/postprocess.py:29-42."""
    return engine.bbox_fill(pred)


def find_postprocessor(key: str) -> Callable[[np.ndarray, SingleData], np.ndarray]:
    return POSTPROCESSORS[key.lower().replace('_', '').replace('-', '')]


def postprocess_help():
    return (
        "Postprocessors available:\n"
        "cc_majority:    classify all pixels of each connected component as most frequent class.\n"
        "bounding_boxes: replace each connected component in the prediction with its bounding box.\n"
    )


POSTPROCESSORS = {
    'ccmajority': vote_connected_component_class,
    'ccvote': vote_connected_component_class,
    'voteconnectedcomponents': vote_connected_component_class,
    'votecomponents': vote_connected_component_class,
    'boundingboxes': add_bounding_boxes,
    'bbox': add_bounding_boxes,
}
