"""CPU suite (-m "not gpu"): the oracle against (a) the vectors generated from the reference's
own importable code (tests/golden/reference_vectors.json), (b) torch CPU convolutions as an
independent implementation, (c) brute-force restatements of the integer post-process."""
import json
import os

import numpy as np
import pytest

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.json")))


def test_preprocess_matches_reference_for_every_byte(oracle_mod):
    want = np.array(GOLD["preprocess_f32_bits"], np.uint32).view(np.float32)
    got = oracle_mod.preprocess(np.arange(256, dtype=np.uint8))
    assert np.array_equal(got, want)
    # float32 division by 255 is what the engine's LUT holds; a multiply by 1/255 is NOT the same
    assert np.array_equal(want, np.arange(256, dtype=np.float32) / np.float32(255.0))
    assert (np.arange(256, dtype=np.float32) * np.float32(1 / 255.0) != want).sum() > 0


def test_bf16_of_reciprocal_multiply_equals_bf16_of_division(oracle_mod):
    """The MFMA first-layer kernel computes bf16(u * (1/255)); after bf16 rounding that equals
    bf16(u / 255) for all 256 byte values."""
    u = np.arange(256, dtype=np.float32)
    a = oracle_mod.round_bf16(u / np.float32(255.0))
    b = oracle_mod.round_bf16(u * np.float32(0.00392156886))
    assert np.array_equal(a, b)


def test_softmax_argmax_match_reference_call(oracle_mod):
    from scipy.special import softmax
    z = np.array(GOLD["softmax_logits_bits"], np.uint32).view(np.float32).reshape(-1, 3)
    want = np.array(GOLD["softmax_probs_bits"], np.uint32).view(np.float32).reshape(-1, 3)
    assert np.array_equal(softmax(z, -1), want)
    assert oracle_mod.argmax(z).tolist() == GOLD["argmax"]      # first maximum wins on ties


def test_round_bf16_is_round_to_nearest_even(oracle_mod):
    import torch
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.standard_normal(10000).astype(np.float32) * 3,
                        np.array([0.0, -0.0, 1.0, 1.00390625, 1.01171875, 65504.0, 1e-30], np.float32)])
    want = torch.from_numpy(x).to(torch.bfloat16).to(torch.float32).numpy()
    assert np.array_equal(oracle_mod.round_bf16(x), want)


@pytest.mark.parametrize("k,stride,cin,cout", [(5, 1, 7, 11), (3, 1, 4, 9), (3, 2, 5, 6), (2, 1, 3, 4), (1, 1, 8, 3)])
def test_conv_against_torch(oracle_mod, k, stride, cin, cout):
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(k * 10 + stride)
    x = rng.standard_normal((18, 22, cin)).astype(np.float32)
    w = (rng.standard_normal((k, k, cin, cout)) * 0.2).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    y = oracle_mod.conv2d(x, w, b, stride=stride, relu=True)
    # TF SAME: pad_total = max((out-1)*s + k - in, 0), before = total // 2 (extra goes after)
    Ho, pt = oracle_mod.same_pad(18, k, stride)
    Wo, pl = oracle_mod.same_pad(22, k, stride)
    th = max((Ho - 1) * stride + k - 18, 0)
    tw = max((Wo - 1) * stride + k - 22, 0)
    xt = F.pad(torch.from_numpy(x).permute(2, 0, 1)[None], (pl, tw - pl, pt, th - pt))
    yt = F.relu(F.conv2d(xt, torch.from_numpy(w).permute(3, 2, 0, 1), torch.from_numpy(b), stride=stride))
    yt = yt[0].permute(1, 2, 0).numpy()
    assert y.shape == yt.shape
    assert np.abs(y - yt).max() < 1e-4


def test_transposed_convs_against_torch(oracle_mod):
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(3)
    x = rng.standard_normal((9, 13, 7)).astype(np.float32)
    K5 = (rng.standard_normal((5, 5, 6, 7)) * 0.2).astype(np.float32)   # Keras (kh,kw,Cout,Cin)
    wc = np.ascontiguousarray(np.transpose(K5[::-1, ::-1], (0, 1, 3, 2)))
    y = oracle_mod.conv2d(x, wc, None)
    yt = F.conv_transpose2d(torch.from_numpy(x).permute(2, 0, 1)[None], torch.from_numpy(K5).permute(3, 2, 0, 1), padding=2)
    assert np.abs(y - yt[0].permute(1, 2, 0).numpy()).max() < 1e-4
    K2 = (rng.standard_normal((2, 2, 6, 7)) * 0.2).astype(np.float32)
    y2 = oracle_mod.deconv2x2(x, np.transpose(K2, (0, 1, 3, 2)), None)
    yt2 = F.conv_transpose2d(torch.from_numpy(x).permute(2, 0, 1)[None], torch.from_numpy(K2).permute(3, 2, 0, 1), stride=2)
    assert np.abs(y2 - yt2[0].permute(1, 2, 0).numpy()).max() < 1e-5


def test_model_shapes_param_counts_and_pad_crop(oracle_mod):
    # fcn_skip C=3 has 673 013 parameters in 26 tensors
    Wt = oracle_mod.init_weights("fcn_skip", 3)
    assert len(Wt) == 26 and sum(v.size for v in Wt.values()) == 673013
    rng = np.random.default_rng(0)
    for arch in oracle_mod.ARCHS:
        Wa = oracle_mod.init_weights(arch, 4, bias_scale=0.05)
        img = rng.integers(0, 256, (37, 45), dtype=np.uint8)        # not a multiple of 32
        z = oracle_mod.forward(arch, Wa, img)
        assert z.shape == (37, 45, 4) and np.isfinite(z).all()
    # the zero-padded canvas is part of the semantics: bottom/right pad influences border pixels,
    # and a page that is already a multiple of 32 equals its own canvas
    img = rng.integers(0, 256, (64, 32), dtype=np.uint8)
    Wt = oracle_mod.init_weights("fcn_skip", 3, bias_scale=0.05)
    z_full = oracle_mod.forward("fcn_skip", Wt, img)
    z_crop = oracle_mod.forward("fcn_skip", Wt, img[:60, :30])
    assert z_crop.shape == (60, 30, 3)
    assert not np.allclose(z_full[:60, :30], z_crop)


def test_fcn_forward_against_torch_graph(oracle_mod):
    """Whole fcn_skip forward rebuilt with torch.nn.functional (independent of the C oracle)."""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(5)
    Wt = oracle_mod.init_weights("fcn_skip", 3, seed=9, gain=1.5, bias_scale=0.05)
    img = rng.integers(0, 256, (64, 96), dtype=np.uint8)
    T = {k: torch.from_numpy(v) for k, v in Wt.items()}

    def conv(x, n, relu):
        y = F.conv2d(x, T[n + "/kernel"].permute(3, 2, 0, 1), T[n + "/bias"], padding=2)
        return F.relu(y) if relu else y

    def tconv5(x, n):
        return F.relu(F.conv_transpose2d(x, T[n + "/kernel"].permute(3, 2, 0, 1), T[n + "/bias"], padding=2))

    def dec2(x, n, relu):
        y = F.conv_transpose2d(x, T[n + "/kernel"].permute(3, 2, 0, 1), T[n + "/bias"], stride=2)
        return F.relu(y) if relu else y

    x = (torch.from_numpy(img.astype(np.float32)) / 255.0)[None, None]
    c1 = conv(x, "conv2d", True); c2 = conv(c1, "conv2d_1", False)
    c3 = conv(F.max_pool2d(c2, 2), "conv2d_2", True); c4 = conv(c3, "conv2d_3", False)
    c5 = conv(F.max_pool2d(c4, 2), "conv2d_4", True); c6 = conv(c5, "conv2d_5", False)
    c7 = conv(F.max_pool2d(c6, 2), "conv2d_6", True)
    d1 = tconv5(c7, "conv2d_transpose")
    d2 = torch.cat([dec2(d1, "conv2d_transpose_1", True), c6], 1)
    d3 = torch.cat([tconv5(d2, "conv2d_transpose_2"), c5], 1)
    d4 = torch.cat([dec2(d3, "conv2d_transpose_3", True), c3], 1)
    d5 = torch.cat([dec2(d4, "conv2d_transpose_4", False), c2], 1)
    zt = F.conv2d(d5, T["logits/kernel"].permute(3, 2, 0, 1), T["logits/bias"])[0].permute(1, 2, 0).numpy()
    z = oracle_mod.forward("fcn_skip", Wt, img)
    assert np.abs(z - zt).max() < 2e-4 * max(1.0, np.abs(zt).max())


def _bruteforce_components(binary):
    H, W = binary.shape
    lab = -np.ones((H, W), np.int64)
    n = 0
    for y in range(H):
        for x in range(W):
            if binary[y, x] and lab[y, x] < 0:
                stack = [(y, x)]
                lab[y, x] = n
                while stack:
                    cy, cx = stack.pop()
                    for dy, dx in ((1, 0), (-1, 0), (0, 1), (0, -1)):
                        ny, nx = cy + dy, cx + dx
                        if 0 <= ny < H and 0 <= nx < W and binary[ny, nx] and lab[ny, nx] < 0:
                            lab[ny, nx] = n
                            stack.append((ny, nx))
                n += 1
    return lab, n


def test_cc_vote_against_bruteforce(oracle_mod):
    rng = np.random.default_rng(2)
    pred = rng.integers(0, 4, (23, 31)).astype(np.int64)
    binary = (rng.random((23, 31)) < 0.45).astype(np.uint8)
    lab, n = _bruteforce_components(binary)
    want = pred.copy()
    for i in range(n):
        m = lab == i
        want[m] = np.argmax(np.bincount(pred[m], minlength=4))   # tie -> lowest class
    assert np.array_equal(oracle_mod.vote_connected_component_class(pred, binary), want)
    # tie example from lib/postprocess.py:22-23 semantics
    p = np.array([[2, 1, 1, 2]], np.int64)
    assert oracle_mod.vote_connected_component_class(p, np.ones((1, 4), np.uint8)).tolist() == [[1, 1, 1, 1]]


def test_bbox_masks_otsu(oracle_mod):
    pred = np.zeros((8, 8), np.int64)
    pred[1, 1] = 1; pred[3, 3] = 1; pred[2:4, 5] = 2; pred[3, 4] = 2
    out = oracle_mod.add_bounding_boxes(pred)
    assert out[1, 1] == 1 and out[3, 3] == 1 and out[2, 2] == 0      # diagonal pixels are separate boxes
    assert out[2, 4] == 2 and out[3, 5] == 2                            # box of the L-shaped class-2 part
    lut = np.array([[0, 0, 0], [255, 0, 0], [0, 255, 0]], np.uint8)
    binary = (np.arange(64).reshape(8, 8) % 2).astype(np.uint8)
    color, overlay, inverted, fg = oracle_mod.generate_output_masks(pred, binary, lut)
    assert np.array_equal(color[1, 1], [255, 0, 0])
    assert (overlay[binary == 1] == 0).all() and np.array_equal(overlay[binary == 0], color[binary == 0])
    assert (inverted[binary == 0] == 0).all() and np.array_equal(inverted, fg)
    g = np.concatenate([np.full(500, 40, np.uint8), np.full(300, 200, np.uint8)]).reshape(20, 40)
    t = oracle_mod.otsu_threshold(g)
    assert 40 <= t < 200




def _round_f32(fr):
    """Correctly rounded float32 of a Fraction (round-to-nearest-even), without an intermediate float64 rounding."""
    from fractions import Fraction
    if fr == 0:
        return np.float32(0.0)
    sign = -1 if fr < 0 else 1
    a = abs(fr)
    import math
    e = math.floor(math.log2(a)) if a >= 1 else -math.ceil(-math.log2(a))
    while Fraction(2) ** e > a:
        e -= 1
    while Fraction(2) ** (e + 1) <= a:
        e += 1
    e = max(e, -126)                              # subnormals share the smallest exponent
    q = a / (Fraction(2) ** (e - 23))             # 24 significant bits in the integer part
    n, rem = divmod(q.numerator, q.denominator)
    twice = 2 * rem
    if twice > q.denominator or (twice == q.denominator and (n & 1)):
        n += 1
    return np.float32(sign * float(Fraction(n) * Fraction(2) ** (e - 23)))


def test_conv_chain_is_blocked_in_slabs_of_16_channels(oracle_mod):
    """Pins the accumulation order the float32 engine and the C oracle share (oracle/pseg_oracle.c header): acc = +0; for
    every slab of 16 input channels: for ky, kx, ci in the slab: acc = fmaf(x, w, acc); out = acc + bias.  Restated here with
    exact rational arithmetic and one correctly rounded float32 per fmaf -- no libm, no C -- and compared bit for bit; the
    all-channels-inside-each-tap order of rounds 1-2 must differ somewhere (the test discriminates)."""
    from fractions import Fraction
    rng = np.random.default_rng(3)
    for Cin, Cout, k in ((20, 3, 3), (40, 2, 3), (16, 2, 5)):
        H, W = 5, 6
        x = rng.standard_normal((H, W, Cin)).astype(np.float32)
        w = (rng.standard_normal((k, k, Cin, Cout)) * 0.3).astype(np.float32)
        b = rng.standard_normal(Cout).astype(np.float32)
        got = oracle_mod.core.conv2d(x, w, b)
        pt = (k - 1) // 2

        def chain(order):
            out = np.zeros((H, W, Cout), np.float32)
            for y in range(H):
                for xx in range(W):
                    for co in range(Cout):
                        acc = np.float32(0.0)
                        for (ky, kx, ci) in order:
                            iy, ix = y + ky - pt, xx + kx - pt
                            if 0 <= iy < H and 0 <= ix < W:
                                acc = _round_f32(Fraction(float(x[iy, ix, ci])) * Fraction(float(w[ky, kx, ci, co])) + Fraction(float(acc)))
                        out[y, xx, co] = np.float32(acc + b[co])
            return out
        blocked = [(ky, kx, ci) for cb in range(0, Cin, 16) for ky in range(k) for kx in range(k) for ci in range(cb, min(cb + 16, Cin))]
        plain = [(ky, kx, ci) for ky in range(k) for kx in range(k) for ci in range(Cin)]
        want = chain(blocked)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (Cin, Cout, k)
        if Cin > 16:
            assert not np.array_equal(chain(plain).view(np.uint32), want.view(np.uint32)), "the two orders agree everywhere: the case does not discriminate"


def test_brightness_shift_oracle_is_the_published_arithmetic():
    """oracle.augment.apply_brightness_shift runs keras-preprocessing 1.1.2's apply_brightness_shift(x, b, scale=False) through the
    installed Pillow; here against the closed form of the same steps (array_to_img's truncation to uint8, ImagingBlend with a black
    image: truncating interpolation for b in [0, 1], clipped truncating extrapolation otherwise, the stretch to 8 bit and back for
    planes outside [0, 255]) -- the form csrc/pseg_resize.hip's kernel follows."""
    from oracle import augment as A

    def closed_form(x, b):
        x = np.asarray(x, np.float32)
        lo, hi = np.min(x), np.max(x)
        local = lo < 0 or hi > 255
        y = x.copy()
        if local:
            y = y - lo
            m = np.max(y)
            if m != 0:
                y /= m
            y *= np.float32(255)
        u = y.astype('uint8').astype(np.float32)
        t = np.float32(b) * u
        w = np.where(t <= 0, 0, np.where(t >= 255, 255, np.floor(t))).astype(np.float32)
        return (w / np.float32(255) * (hi - lo) + lo).astype(np.float32) if local else w

    rng = np.random.default_rng(0)
    for scale, shift in ((255, 0), (280, 10), (200, 60), (1, -93), (262, 3)):
        x = (rng.random((31, 41, 1)) * scale - shift).astype(np.float32)
        for b in (0.0, 0.37, 0.8, 1.0, 1.2, 1.9, 3.5):
            assert np.array_equal(A.apply_brightness_shift(x, b), closed_form(x, b)), (scale, shift, b)
