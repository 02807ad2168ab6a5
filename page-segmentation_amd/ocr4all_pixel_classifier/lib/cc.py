"""cc_bbox / cc_bbox_func (reference: lib/cc.py): views of a component's bounding box from a cv2-style stats
table (columns LEFT, TOP, WIDTH, HEIGHT, AREA)."""
import numpy as np

CC_STAT_LEFT, CC_STAT_TOP, CC_STAT_WIDTH, CC_STAT_HEIGHT, CC_STAT_AREA = range(5)


def cc_bbox(image: np.ndarray, cc_stats, cc_index):
    return cc_bbox_func(cc_stats, cc_index)(image)


def cc_bbox_func(cc_stats, cc_index):
    left, top = cc_stats[cc_index, CC_STAT_LEFT], cc_stats[cc_index, CC_STAT_TOP]
    w, h = cc_stats[cc_index, CC_STAT_WIDTH], cc_stats[cc_index, CC_STAT_HEIGHT]
    return lambda image: image[top:top + h, left:left + w]
