"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's evaluation helpers (lib/image_ops.py:8-55,
lib/evaluation.py:8-117, lib/cc.py) in NumPy, with scipy.ndimage.label standing in for
cv2.connectedComponentsWithStats (OpenCV is absent offline: parity unpinned for the component numbering --
raster order of the first pixel for connectivity 4, of the first 2x2 block for connectivity 8, which is the order
OpenCV's SAUF / Spaghetti scans create provisional labels in)."""
import numpy as np
from scipy import ndimage


def fgpa(pred, mask, bin):                                   # lib/image_ops.py:8-19
    pfg = pred * bin
    mfg = mask * bin
    fg_count = np.count_nonzero(bin)
    return (fg_count - np.count_nonzero(pfg != mfg)) / fg_count


def fgoverlap_per_class(pred, mask, bin, n_classes):         # lib/image_ops.py:22-55
    pfg = (pred.astype(np.int64) + 1) * bin - 1
    mfg = (mask.astype(np.int64) + 1) * bin - 1
    rows = []
    for i in range(n_classes + 1):
        actual, expected = (pfg == i).astype(np.uint8), (mfg == i).astype(np.uint8)
        poi = actual + expected
        if np.count_nonzero(poi) == 0:
            rows.append((np.nan, 0, 0, 0))
            continue
        fp = np.count_nonzero(actual > expected)
        fn = np.count_nonzero(expected > actual)
        tp = np.count_nonzero(poi == 2)
        rows.append((tp / (tp + fp + fn), tp, fp, fn))
    return tuple(map(list, zip(*rows)))


def count_matches(mask, pred, label):                        # lib/evaluation.py:8-22
    ml, pl = mask == label, pred == label
    return (np.count_nonzero(ml & pl), np.count_nonzero(ml & ~pl), np.count_nonzero(~ml & pl))


def total_accuracy(mask, pred):                              # lib/evaluation.py:25-33
    eq = mask == pred
    return np.count_nonzero(eq), eq.size


def connected_components_with_stats(binary, connectivity=4):
    """(num_labels, labels int32, stats (N,5) int32 [LEFT, TOP, WIDTH, HEIGHT, AREA], centroids (N,2) float64)."""
    b = np.asarray(binary) != 0
    H, W = b.shape
    st = ndimage.generate_binary_structure(2, 1 if connectivity == 4 else 2)
    lab, n = ndimage.label(b, structure=st)
    if n:
        ys, xs = np.nonzero(lab)
        ids = lab[ys, xs]
        key = ys * W + xs if connectivity == 4 else (ys >> 1) * ((W + 1) >> 1) + (xs >> 1)
        first = np.full(n + 1, np.iinfo(np.int64).max, np.int64)
        np.minimum.at(first, ids, key)
        rank = np.empty(n + 1, np.int64)
        rank[0] = 0
        rank[1:][np.argsort(first[1:], kind="stable")] = np.arange(1, n + 1)
        lab = rank[lab]
    lab = lab.astype(np.int32)
    stats = np.zeros((n + 1, 5), np.int32)
    cent = np.full((n + 1, 2), np.nan)
    for i in range(n + 1):
        ys, xs = np.nonzero(lab == i)
        if len(ys) == 0:
            continue
        stats[i] = (xs.min(), ys.min(), xs.max() - xs.min() + 1, ys.max() - ys.min() + 1, len(ys))
        cent[i] = (xs.sum() / len(xs), ys.sum() / len(ys))
    return n + 1, lab, stats, cent


def cc_equal(threshold):                                     # lib/evaluation.py:52-53
    return lambda pred, mask: np.count_nonzero(pred == mask) / np.size(mask) >= threshold


def cc_matching(label, threshold_tp, threshold_fp, threshold_mask=None):     # lib/evaluation.py:56-70
    if not threshold_mask:
        threshold_mask = threshold_tp

    def match(mask, pred):
        size = np.size(mask)
        pfp = np.count_nonzero(pred == label) / size >= threshold_fp
        ptp = np.count_nonzero(pred == label) / size >= threshold_tp
        mm = np.count_nonzero(mask == label) / size >= threshold_mask
        return np.array([int(ptp and mm), int(pfp and not mm), int(mm and not ptp)])
    return match


def run_per_component(mask, pred, binary, func, connectivity=4, only_label=None, threshold=None):
    """ConnectedComponentEval(...).only_label(...).run_per_component(func) as a list (lib/evaluation.py:73-117)."""
    n, lab, stats, _ = connected_components_with_stats(binary.astype("uint8"), connectivity)
    out = []
    for i in range(1, n):
        l, t, w, h = stats[i, :4]
        box = lambda im: im[t:t + h, l:l + w]
        sel = box(lab) == i
        if only_label:
            def ratio(img):
                px = box(img)[sel]
                return np.count_nonzero(px == only_label) / np.size(px)
            if not (ratio(mask) >= threshold or ratio(pred) > 0):
                continue
        out.append(func(box(mask)[sel], box(pred)[sel]))
    return out
