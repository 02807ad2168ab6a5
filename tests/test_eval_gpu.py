"""Evaluation reductions on the GPU (lib/image_ops.py:8-55, lib/evaluation.py) against the NumPy /
scipy restatement: integer counts bit-exact, ratios equal as Python floats."""
import numpy as np
import pytest

from oracle import evaluation as O

pytestmark = pytest.mark.gpu


def _page(seed, H, W, C):
    from pseg_amd import synth
    img, binary, mask = synth.synth_page(seed, max(H, 96), max(W, 96), C)          # the generator needs room for its layout
    binary, mask = np.ascontiguousarray(binary[:H, :W]), np.ascontiguousarray(mask[:H, :W])
    rng = np.random.default_rng(seed)
    pred = mask.astype(np.int64)
    flip = rng.random(mask.shape) < 0.07
    pred[flip] = rng.integers(0, C, int(flip.sum()))
    return binary, mask, pred


@pytest.mark.parametrize("shape,C", [((96, 128), 3), ((257, 131), 6), ((1, 1), 3), ((64, 1), 3), ((1, 200), 4)])
def test_pixel_metrics(gpu, shape, C):
    from ocr4all_pixel_classifier.lib import image_ops as I
    from ocr4all_pixel_classifier.lib import evaluation as E
    binary, mask, pred = _page(11, shape[0], shape[1], C)
    if binary.sum() == 0:
        binary = np.ones_like(binary)
    assert I.fgpa(pred, mask, binary) == O.fgpa(pred, mask, binary)
    got, want = I.fgoverlap_per_class(pred, mask, binary, C), O.fgoverlap_per_class(pred, mask, binary, C)
    for g, w in zip(got, want):
        assert len(g) == C + 1 and all((a == b) or (np.isnan(a) and np.isnan(b)) for a, b in zip(g, w))
    for label in range(C + 1):
        assert E.count_matches(mask, pred, label) == O.count_matches(mask, pred, label)
    assert E.total_accuracy(mask, pred) == O.total_accuracy(mask, pred)
    assert E.f1_measures(0, 3, 4) == (0.0, 0.0, 0.0) and E.f1_measures(6, 2, 6) == (0.75, 0.5, 0.6)
    # uint8 predictions (the engine's compact label map) give the same counts
    assert E.count_matches(mask, pred.astype(np.uint8), 1) == O.count_matches(mask, pred, 1)


def test_confusion_histogram_layout_and_wide_class_range(gpu):
    rng = np.random.default_rng(0)
    for C in (2, 40, 200):                                      # 200: (C+1)^2 * 2 slots exceed the LDS histogram
        pred = rng.integers(-1, C + 2, (50, 70)).astype(np.int64)
        mask = rng.integers(0, C, (50, 70)).astype(np.int32)
        b = rng.integers(0, 2, (50, 70)).astype(np.uint8) * 255
        got = gpu.eval_confusion(pred, mask, b, C)
        want = np.zeros((2, C + 1, C + 1), np.int64)
        pc = np.where((pred < 0) | (pred >= C), C, pred)
        np.add.at(want, ((b != 0).astype(int), mask, pc), 1)
        assert np.array_equal(got, want)
    assert gpu.eval_confusion(np.zeros((0, 5), np.int64), np.zeros((0, 5), np.uint8), None, 3).sum() == 0
    with pytest.raises(gpu.PsegError):
        gpu.eval_confusion(np.zeros((2, 2), np.int64), np.zeros((2, 3), np.uint8), None, 3)


@pytest.mark.parametrize("conn", [4, 8])
@pytest.mark.parametrize("shape", [(96, 128), (257, 131), (1, 70), (70, 1), (2, 2)])
def test_cc_label_stats_match_restatement(gpu, conn, shape):
    binary, mask, pred = _page(5, shape[0], shape[1], 3)
    rng = np.random.default_rng(shape[0])
    binary = (binary | (rng.random(shape) < 0.08)).astype(np.uint8)       # specks: many small components, diagonals
    n, lab = gpu.cc_label(binary, conn)
    wn, wl, ws, wc = O.connected_components_with_stats(binary, conn)
    assert n == wn and lab.dtype == np.int32 and np.array_equal(lab, wl)
    t = gpu.cc_tables(lab, n, pred, mask, 3, want_order=True)
    assert np.array_equal(t["stats"], ws)
    assert np.array_equal(np.isnan(t["centroids"]), np.isnan(wc)) and np.array_equal(np.nan_to_num(t["centroids"]), np.nan_to_num(wc))
    order = t["order"]
    assert np.array_equal(np.sort(order), np.arange(lab.size))
    assert np.array_equal(lab.ravel()[order], np.sort(lab.ravel(), kind="stable"))
    off = np.concatenate([[0], np.cumsum(ws[:, 4])])
    for i in range(n):
        o = order[off[i]:off[i + 1]]
        assert (np.diff(o) > 0).all()                               # raster order inside a component
        sel = lab == i
        assert t["eq"][i] == np.count_nonzero(pred[sel] == mask[sel])
        assert np.array_equal(t["hist_pred"][i, :3], np.bincount(pred[sel], minlength=3))
        assert np.array_equal(t["hist_mask"][i, :3], np.bincount(mask[sel], minlength=3))


def test_all_ink_and_all_paper(gpu):
    for fill in (0, 1):
        b = np.full((40, 50), fill, np.uint8)
        n, lab = gpu.cc_label(b, 4)
        assert n == 1 + fill and (lab == fill).all()
        t = gpu.cc_tables(lab, n)
        assert t["stats"][fill].tolist() == [0, 0, 50, 40, 2000]


@pytest.mark.parametrize("conn", [4, 8])
def test_connected_component_eval_class(gpu, conn):
    from ocr4all_pixel_classifier.lib import evaluation as E
    binary, mask, pred = _page(9, 160, 192, 3)
    ev = E.ConnectedComponentEval(mask, pred, binary, connectivity=conn)
    wn, wl, ws, wc = O.connected_components_with_stats(binary, conn)
    assert ev.num_labels == wn and np.array_equal(ev.labels, wl) and np.array_equal(ev.stats, ws)
    # the two matchers of the reference: answered from the GPU tables
    assert list(ev.run_per_component(E.cc_equal(0.9))) == O.run_per_component(mask, pred, binary, O.cc_equal(0.9), conn)
    got = list(ev.run_per_component(E.cc_matching(1, 0.5, 0.1)))
    want = O.run_per_component(mask, pred, binary, O.cc_matching(1, 0.5, 0.1), conn)
    assert len(got) == len(want) and all(np.array_equal(g, w) for g, w in zip(got, want))
    # an arbitrary callback sees the same pixel slices in the same order
    f = lambda m, p: (int(m.sum()), int(p[0]), int(p[-1]), m.size)
    assert list(ev.run_per_component(f)) == O.run_per_component(mask, pred, binary, f, conn)
    # only_label filter
    ev.only_label(2, 0.5)
    assert list(ev.run_per_component(E.cc_equal(0.5))) == O.run_per_component(mask, pred, binary, O.cc_equal(0.5), conn, 2, 0.5)
    assert list(ev.run_per_component(f)) == O.run_per_component(mask, pred, binary, f, conn, 2, 0.5)
    with pytest.raises(ValueError):
        E.ConnectedComponentEval(mask, pred, np.zeros((4, 4, 3), np.uint8))


def test_full_page_counts_consistency(gpu):
    """BASELINE configs[4] size: identities that hold at any size (the restatement loops per component)."""
    binary, mask, pred = _page(2, 4096, 3072, 6)
    c = gpu.eval_confusion(pred, mask, binary, 6)
    assert c.sum() == 4096 * 3072 and c[1].sum() == np.count_nonzero(binary)
    assert np.trace(c[0]) + np.trace(c[1]) == np.count_nonzero(pred == mask)
    n, lab = gpu.cc_label(binary, 4)
    assert np.array_equal(lab != 0, binary != 0)
    t = gpu.cc_tables(lab, n, pred, mask, 6)
    assert t["stats"][:, 4].sum() == lab.size and t["eq"].sum() == np.count_nonzero(pred == mask)
    assert np.array_equal(t["hist_pred"][1:].sum(0)[:6], np.bincount(pred[binary != 0], minlength=6))
    first = np.full(n, lab.size, np.int64)                              # numbering = raster order of first pixels
    np.minimum.at(first, lab.ravel(), np.arange(lab.size))
    assert (np.diff(first[1:]) > 0).all()


def test_cc_label_random_shapes_and_densities(gpu):
    """Tile-local labelling + border unions (16 x 64 tiles): shapes around the tile sizes and ink densities from
    isolated specks to percolating blobs, both connectivities, against the scipy restatement (same numbering)."""
    rng = np.random.default_rng(123)
    shapes = [(15, 63), (16, 64), (17, 65), (31, 129), (48, 200), (100, 64), (33, 300), (130, 70)]
    for (H, W) in shapes:
        for dens in (0.05, 0.35, 0.5, 0.62, 0.9):
            b = (rng.random((H, W)) < dens).astype(np.uint8)
            for conn in (4, 8):
                n, lab = gpu.cc_label(b, conn)
                wn, wl, _, _ = O.connected_components_with_stats(b, conn)
                assert n == wn and np.array_equal(lab, wl), (H, W, dens, conn)
    # long thin structures crossing many tiles: a spiral and a comb
    b = np.zeros((70, 200), np.uint8)
    b[::2, :] = 1
    b[1::4, -1] = 1
    b[3::4, 0] = 1                                             # one serpentine component
    for conn in (4, 8):
        n, lab = gpu.cc_label(b, conn)
        wn, wl, _, _ = O.connected_components_with_stats(b, conn)
        assert n == wn == 2 and np.array_equal(lab, wl)
