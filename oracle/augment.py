"""Augmentation warp reference (TEST INFRASTRUCTURE ONLY): the affine warp keras-preprocessing 1.1.2 applies per
channel, scipy.ndimage.affine_transform(x, matrix, offset, order, mode='nearest', cval) -- here with the scipy
installed beside the tests (the reference's pinned scipy / keras-preprocessing are absent offline: PARITY
UNPINNED; the cubic B-spline prefilter boundary handling changed in scipy 1.6, results away from the border
agree to float32 rounding)."""
import numpy as np
from scipy import ndimage


def affine_transform(plane, matrix, offset, order):
    return ndimage.affine_transform(np.asarray(plane, np.float32), np.asarray(matrix, np.float64),
                                    np.asarray(offset, np.float64), order=order, mode='nearest', cval=0.0)
