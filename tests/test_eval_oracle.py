"""oracle/evaluation.py against hand-computed cases (the reference holds no fixtures for these helpers; cv2 is
absent offline, so the component numbering is pinned only by these cases and OpenCV's documented scan order)."""
import numpy as np

from oracle import evaluation as E


def test_fgpa_and_overlap_small_case():
    pred = np.array([[0, 1, 1], [2, 2, 0]], np.uint8)
    mask = np.array([[0, 1, 2], [2, 1, 1]], np.uint8)
    bin_ = np.array([[1, 1, 1], [1, 1, 0]], np.uint8)
    assert E.fgpa(pred, mask, bin_) == 3 / 5
    ov, tp, fp, fn = E.fgoverlap_per_class(pred, mask, bin_, 3)
    assert tp == [1, 1, 1, 0] and fp == [0, 1, 1, 0] and fn == [0, 1, 1, 0]
    assert ov[0] == 1.0 and ov[1] == 1 / 3 and ov[2] == 1 / 3 and np.isnan(ov[3])
    assert E.count_matches(mask, pred, 1) == (1, 2, 1)
    assert E.total_accuracy(mask, pred) == (3, 6)


def test_component_numbering_and_stats():
    b = np.zeros((6, 8), np.uint8)
    b[1, 6] = 1                      # A: first in raster order, block (0, 3)
    b[1, 0] = 1; b[2, 1] = 1         # B: diagonal pair: one component for 8, two for 4
    b[4:6, 3:6] = 1                  # C
    n4, l4, s4, c4 = E.connected_components_with_stats(b, 4)
    assert n4 == 5 and l4[1, 0] == 1 and l4[1, 6] == 2 and l4[2, 1] == 3 and l4[4, 3] == 4
    assert s4[4].tolist() == [3, 4, 3, 2, 6] and c4[4].tolist() == [4.0, 4.5]
    assert s4[0, 4] == 48 - 9
    n8, l8, s8, _ = E.connected_components_with_stats(b, 8)
    assert n8 == 4 and l8[1, 0] == l8[2, 1] == 1 and l8[1, 6] == 2 and l8[4, 4] == 3
    # block order differs from raster order: X starts on row 1 (block row 0), Y on row 0 further right
    b2 = np.zeros((4, 8), np.uint8)
    b2[1, 0] = 1
    b2[0, 5] = 1
    _, l, _, _ = E.connected_components_with_stats(b2, 8)
    assert l[1, 0] == 1 and l[0, 5] == 2
    _, l, _, _ = E.connected_components_with_stats(b2, 4)
    assert l[0, 5] == 1 and l[1, 0] == 2


def test_run_per_component_matching():
    b = np.zeros((4, 6), np.uint8)
    b[0, 0:2] = 1
    b[2:4, 3:6] = 1
    mask = np.where(b, 1, 0).astype(np.uint8)
    pred = mask.copy()
    pred[3, 5] = 2
    assert E.run_per_component(mask, pred, b, E.cc_equal(0.9)) == [True, False]
    got = E.run_per_component(mask, pred, b, E.cc_matching(1, 0.5, 0.1))
    assert [g.tolist() for g in got] == [[1, 0, 0], [1, 0, 0]]
    assert E.run_per_component(mask, pred, b, E.cc_equal(0.5), only_label=2, threshold=0.5) == [True]
