"""Shape helpers (reference: lib/util.py)."""
import numpy as np


def gray_to_rgb(img):
    """lib/util.py:4-9: (H,W) -> (H,W,3) by replication; RGB input passes through."""
    if img.ndim == 3 and img.shape[2] == 3:
        return img
    return np.repeat(img[..., None], 3, axis=-1)


def image_to_batch(img):
    """lib/util.py:12-18: (H,W) -> (1,H,W,1); (H,W,C) -> (1,H,W,C)."""
    if img.ndim == 2:
        return img[None, :, :, None]
    return img[None]


def preserving_resize(image, target_shape):
    """lib/util.py:21-29: nearest-neighbour resize that keeps values (order 0, no anti-aliasing,
    preserve_range).  Index arithmetic only (a gather), float64 result like scikit-image's."""
    image = np.asarray(image)
    H, W = image.shape[:2]
    Ho, Wo = int(target_shape[0]), int(target_shape[1])
    r = np.floor((np.arange(Ho) + 0.5) * (H / Ho)).astype(np.int64).clip(0, H - 1)
    c = np.floor((np.arange(Wo) + 0.5) * (W / Wo)).astype(np.int64).clip(0, W - 1)
    return image[r][:, c].astype(np.float64)
