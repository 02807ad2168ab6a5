"""Integer post-process kernels (CC vote, bbox fill, masks, Otsu/char height) vs the oracle:
bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _case(rng, H, W, C, density=0.35):
    pred = rng.integers(0, C, size=(H, W)).astype(np.int64)
    # blocky class regions + noisy binary with blobs
    yy, xx = np.mgrid[0:H, 0:W]
    pred = ((yy // 7 + xx // 11 + pred // max(C - 1, 1)) % C).astype(np.int64)
    binary = (rng.random((H, W)) < density).astype(np.uint8)
    return pred, binary


@pytest.mark.parametrize("H,W,C", [(1, 1, 3), (5, 64, 3), (64, 5, 6), (97, 131, 3), (256, 192, 6), (300, 517, 4)])
def test_cc_vote(gpu, oracle_mod, H, W, C):
    rng = np.random.default_rng(H * 1000 + W)
    pred, binary = _case(rng, H, W, C)
    want = oracle_mod.vote_connected_component_class(pred, binary)
    got = gpu.cc_vote(pred.copy(), binary, C)
    assert np.array_equal(got, want)


def test_cc_vote_ties_and_edges(gpu, oracle_mod):
    # tie -> lowest class; all-ink; no ink; binary with values > 1 (non-zero = ink for cv2)
    pred = np.array([[2, 1, 1, 2], [0, 0, 0, 0]], np.int64)
    binary = np.array([[1, 1, 1, 1], [0, 0, 0, 0]], np.uint8)
    assert np.array_equal(gpu.cc_vote(pred.copy(), binary, 3), oracle_mod.vote_connected_component_class(pred, binary))
    assert gpu.cc_vote(pred.copy(), binary, 3)[0].tolist() == [1, 1, 1, 1]
    ones = np.ones((40, 70), np.uint8)
    p2 = np.random.default_rng(0).integers(0, 5, (40, 70)).astype(np.int64)
    assert np.array_equal(gpu.cc_vote(p2.copy(), ones, 5), oracle_mod.vote_connected_component_class(p2, ones))
    assert np.array_equal(gpu.cc_vote(p2.copy(), ones * 0, 5), p2)
    assert np.array_equal(gpu.cc_vote(p2.copy(), ones * 255, 5), oracle_mod.vote_connected_component_class(p2, ones * 255))
    # spiral / snake component stresses the union-find
    snake = np.zeros((65, 65), np.uint8)
    snake[::2, :] = 1
    snake[1::4, -1] = 1
    snake[3::4, 0] = 1
    p3 = np.ascontiguousarray(np.tile(p2, (2, 1))[:65, :65])
    assert np.array_equal(gpu.cc_vote(p3.copy(), snake, 5), oracle_mod.vote_connected_component_class(p3, snake))


@pytest.mark.parametrize("H,W,C", [(1, 1, 2), (37, 64, 3), (128, 97, 6)])
def test_bbox_fill(gpu, oracle_mod, H, W, C):
    rng = np.random.default_rng(H + W)
    pred = np.zeros((H, W), np.int64)
    for _ in range(12):
        y, x = rng.integers(0, H), rng.integers(0, W)
        h, w = rng.integers(1, max(2, H // 3)), rng.integers(1, max(2, W // 3))
        pred[y:y + h, x:x + w] = rng.integers(0, C)
    pred[rng.random((H, W)) < 0.02] = C - 1
    assert np.array_equal(gpu.bbox_fill(pred, C), oracle_mod.add_bounding_boxes(pred))


@pytest.mark.parametrize("H,W", [(1, 1), (3, 5), (64, 64), (97, 131)])
def test_masks(gpu, oracle_mod, H, W):
    rng = np.random.default_rng(H * W)
    pred = rng.integers(0, 6, (H, W)).astype(np.int64)
    binary = rng.integers(0, 2, (H, W)).astype(np.uint8)
    lut = rng.integers(0, 256, (6, 3)).astype(np.uint8)
    want = oracle_mod.generate_output_masks(pred, binary, lut)
    got = gpu.masks(pred, binary, lut)
    for g, w_ in zip(got, want):
        assert np.array_equal(g, w_)


def test_otsu_char_height(gpu, oracle_mod):
    import pseg_amd.synth as synth
    img, binary, mask = synth.synth_page(0, 512, 512)
    gray = 255 - img                      # the un-inverted scan
    h, t = gpu.otsu_char_height(gray, inverse=False)
    assert t == oracle_mod.otsu_threshold(gray)
    assert h == oracle_mod.compute_char_height_from_gray(gray, inverse=False)
    flat = np.full((64, 64), 200, np.uint8)
    h2, _ = gpu.otsu_char_height(flat)
    assert h2 is None and oracle_mod.compute_char_height_from_gray(flat) is None


def test_post_process_call_sequences(gpu, oracle_mod):
    """The cached vote workspace and the resize scratch must not leak between calls of different sizes."""
    rng = np.random.default_rng(23)
    cases = []
    for H, W, C in [(300, 517, 4), (64, 5, 6), (1200, 900, 3), (97, 131, 3), (1, 1, 3)]:
        pred, binary = _case(rng, H, W, C)
        cases.append((pred, binary, C, oracle_mod.vote_connected_component_class(pred, binary)))
    for k in rng.integers(0, len(cases), 25):
        pred, binary, C, want = cases[int(k)]
        assert np.array_equal(gpu.cc_vote(pred.copy(), binary, C), want)


def test_cc_vote_random_shapes_densities_and_class_counts(gpu, oracle_mod):
    """The vote's tile pass (32 x 64 tiles labelled and counted in LDS, closed components finished there, open ones through the
    border unions, the root merge and the run list; more than 12 classes: the page-global labelling with the 32 x 32 counting
    tiles) on shapes around the tile sizes, densities from specks to one percolating component, 2 .. 40 classes, noisy
    predictions."""
    rng = np.random.default_rng(99)
    for (H, W) in [(31, 33), (32, 32), (33, 65), (64, 129), (100, 100), (17, 300)]:
        for dens, C in [(0.1, 2), (0.45, 3), (0.6, 7), (0.95, 40)]:
            binary = (rng.random((H, W)) < dens).astype(np.uint8)
            pred = rng.integers(0, C, size=(H, W)).astype(np.int64)          # every pixel its own class: many keys per tile
            want = oracle_mod.vote_connected_component_class(pred, binary)
            got = gpu.cc_vote(pred.copy(), binary, C)
            assert np.array_equal(got, want), (H, W, dens, C)


def _vote_u8_device(gpu, pred_u8, binary, C):
    """pseg_cc_vote_device_u8 on device buffers (the entry the Predictor chain and bench.py's configs[4] leg use)."""
    import ctypes
    import torch
    from pseg_amd import engine as E
    dev = torch.device("cuda:0")
    d_p = torch.from_numpy(pred_u8.copy()).to(dev)
    d_b = torch.from_numpy(binary).to(dev)
    H, W = pred_u8.shape
    st = torch.cuda.current_stream(dev).cuda_stream
    vp = ctypes.c_void_p
    E._check(E.lib().pseg_cc_vote_device_u8(0, vp(d_p.data_ptr()), vp(d_b.data_ptr()), H, W, C, vp(st)))
    torch.cuda.synchronize()
    return d_p.cpu().numpy()


def test_cc_vote_uint8_and_int64_maps_on_shapes_around_the_tile_edges(gpu, oracle_mod):
    """The vote on uint8 (device entry) and int64 maps gives the oracle's map bit for bit: shapes around the tile edges
    (32 x 64; pages narrower than a tile, widths that are no multiple of four: the byte-load path), specks to one percolating
    component, long thin components across many tiles."""
    rng = np.random.default_rng(5)
    cases = []
    for (H, W) in [(1, 1), (32, 64), (33, 65), (31, 63), (64, 128), (65, 200), (200, 131), (300, 517), (96, 192), (70, 132), (129, 256)]:
        for dens, C in [(0.05, 3), (0.4, 6), (0.55, 8), (0.97, 2), (0.3, 12)]:
            binary = (rng.random((H, W)) < dens).astype(np.uint8)
            pred = rng.integers(0, C, size=(H, W)).astype(np.int64)
            cases.append((pred, binary, C))
    # horizontal and vertical bars crossing every tile, a frame, a comb
    H, W = 130, 260
    binary = np.zeros((H, W), np.uint8)
    binary[5, :] = 1; binary[:, 7] = 1; binary[40, 3:250] = 1; binary[64:66, :] = 1; binary[:, 128] = 1
    binary[100:120, ::2] = 1; binary[120, :] = 1
    pred = ((np.add.outer(np.arange(H), np.arange(W)) // 9) % 5).astype(np.int64)
    cases.append((pred, binary, 5))
    pred2 = pred.copy(); pred2[::7, ::5] = 7
    cases.append((np.minimum(pred2, 7), binary, 8))
    for pred, binary, C in cases:
        want = oracle_mod.vote_connected_component_class(pred, binary)
        assert np.array_equal(gpu.cc_vote(pred.copy(), binary, C), want), (pred.shape, C)
        got8 = _vote_u8_device(gpu, pred.astype(np.uint8), binary, C)
        assert np.array_equal(got8.astype(np.int64), want), ("u8", pred.shape, C)


def test_cc_vote_structured_extremes_of_the_run_lists(gpu, oracle_mod):
    """The tile pass keeps runs, root slots and border-union tasks in fixed-size per-tile lists: the patterns that fill them -- a
    checkerboard (every other pixel a run of its own: 1 024 runs in a 32 x 64 tile), all ink (one component through every tile
    edge), single-pixel rows / columns (a run per row; a task per row across the left edge), a comb, a spiral through many tiles,
    diagonal stripes -- against the oracle, uint8 and int64 maps, 3 and 12 classes."""
    rng = np.random.default_rng(17)
    H, W = 131, 262
    yy, xx = np.mgrid[0:H, 0:W]
    pats = {
        "checkerboard": ((yy + xx) & 1),
        "all_ink": np.ones((H, W), int),
        "rows": (yy & 1) * np.ones((H, W), int),
        "columns": (xx & 1) * np.ones((H, W), int),
        "comb": ((yy == 3) | ((xx % 3 == 0) & (yy > 3))),
        "diagonals": ((yy + xx) % 4 < 2),
        "border_pairs": (((xx % 64) >= 62) | ((xx % 64) <= 1) | ((yy % 32) == 31) | ((yy % 32) == 0)) & (((yy // 2 + xx // 2) & 1) == 0),
    }
    sp = np.zeros((H, W), int)
    t, l, b, r = 0, 0, H - 1, W - 1
    while t <= b and l <= r:                      # a spiral with gaps of one pixel: one long component
        sp[t, l:r + 1] = 1; sp[t:b + 1, r] = 1
        if t + 2 <= b: sp[b, l + 2:r + 1] = 1
        if l + 2 <= r and t + 2 <= b: sp[t + 2:b + 1, l + 2] = 1
        t += 2; l += 2; b -= 2; r -= 2
        if t <= b and l <= r: sp[t, l] = 1
    pats["spiral"] = sp
    for name, pat in pats.items():
        binary = pat.astype(np.uint8)
        for C in (3, 12):
            pred = rng.integers(0, C, size=(H, W)).astype(np.int64)
            want = oracle_mod.vote_connected_component_class(pred, binary)
            assert np.array_equal(gpu.cc_vote(pred.copy(), binary, C), want), (name, C)
            assert np.array_equal(_vote_u8_device(gpu, pred.astype(np.uint8), binary, C).astype(np.int64), want), (name, C, "u8")


def test_cc_vote_full_size_page_properties(gpu):
    """configs[4] size (4096 x 3072, 6 classes), where the oracle is too slow: the vote is idempotent, leaves paper pixels
    alone, makes every ink run of a row uniform, and the uint8 and int64 entries agree."""
    from pseg_amd import synth
    _, binary, mask = synth.synth_page(1000, 4096, 3072, 6)
    rng = np.random.default_rng(0)
    pred = np.where(rng.random(mask.shape) < 0.2, rng.integers(0, 6, mask.shape), mask).astype(np.uint8)
    voted = _vote_u8_device(gpu, pred, binary, 6)
    assert np.array_equal(_vote_u8_device(gpu, voted, binary, 6), voted)
    assert np.array_equal(voted[binary == 0], pred[binary == 0])
    same_run = (binary[:, 1:] != 0) & (binary[:, :-1] != 0)
    assert np.array_equal(voted[:, 1:][same_run], voted[:, :-1][same_run])
    assert np.array_equal(gpu.cc_vote(pred.astype(np.int64), binary, 6), voted.astype(np.int64))
