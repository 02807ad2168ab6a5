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


def affine_transform_mode(plane, matrix, offset, order, mode, cval=0.0):
    """... with the other fill modes keras-preprocessing accepts ('constant', 'reflect', 'wrap'), float64 planes as scipy is called there."""
    return ndimage.affine_transform(np.asarray(plane, np.float64), np.asarray(matrix, np.float64), np.asarray(offset, np.float64),
                                    order=order, mode=mode, cval=cval).astype(np.float32)


def apply_brightness_shift(x, brightness):
    """keras-preprocessing 1.1.2 affine_transformations.apply_brightness_shift(x, brightness, scale=False) -- the call
    ImageDataGenerator.apply_transform makes -- restated with the installed Pillow (the package itself is absent offline: PARITY
    UNPINNED against it; the arithmetic is array_to_img / ImageEnhance.Brightness / img_to_array as published):
        x_min, x_max = min(x), max(x); local_scale = x_min < 0 or x_max > 255
        img = array_to_img(x, scale=local_scale)       # scale: (x - min) / max * 255; then astype('uint8'), mode 'L' / 'RGB'
        img = ImageEnhance.Brightness(img).enhance(brightness)
        x = img_to_array(img)                          # float32
        if local_scale: x = x / 255 * (x_max - x_min) + x_min"""
    from PIL import Image, ImageEnhance
    x = np.asarray(x, np.float32)
    assert x.ndim == 3 and x.shape[2] in (1, 3)
    x_min, x_max = np.min(x), np.max(x)
    local = bool(x_min < 0 or x_max > 255)
    y = np.array(x, np.float32)
    if local:
        y = y - np.min(y)
        m = np.max(y)
        if m != 0:
            y /= m
        y *= 255
    img = Image.fromarray(y[:, :, 0].astype('uint8'), 'L') if x.shape[2] == 1 else Image.fromarray(y.astype('uint8'), 'RGB')
    img = ImageEnhance.Brightness(img).enhance(brightness)
    out = np.asarray(img, dtype=np.float32)
    if out.ndim == 2:
        out = out[:, :, None]
    if local:
        out = out / 255 * (x_max - x_min) + x_min
    return out.astype(np.float32)
