"""Label-exact throughput mode (north_star: "label maps bit-identical to the CPU reference"; reference argmax:
lib/network.py:259): bf16 pass + margin map + float32 referee on the blocks that hold near-ties must return the
float32 engine's label map EXACTLY (np.array_equal), for random and for trained weights, at BASELINE.json's page size
and on small / odd pages, every graph.  The float32 engine itself is pinned to the oracle bit for bit
(tests/test_predict_gpu.py), so equality with it is equality with the CPU restatement."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    return torch


def _exact_vs_f32(gpu, arch, C, Wt, img, expect_partial=None):
    torch = _torch()
    H, W = img.shape
    dev = torch.device("cuda:0")
    eb = gpu.Engine(arch, C, mode=gpu.MODE_BF16)
    eb.set_weights(Wt)
    e32 = gpu.Engine(arch, C, mode=gpu.MODE_F32_EXACT)
    e32.set_weights(Wt)
    d_img = torch.from_numpy(img).to(dev)
    lab = torch.empty((H, W), dtype=torch.uint8, device=dev)
    lab64 = torch.empty((H, W), dtype=torch.int64, device=dev)
    margin = torch.empty((H, W), dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    eb.predict_exact_labels_device(d_img.data_ptr(), H, W, lab.data_ptr(), d_labels=lab64.data_ptr(), d_margin=margin.data_ptr(), stream=st)
    torch.cuda.synchronize()
    stats = eb.label_exact_stats()
    l32 = torch.empty((H, W), dtype=torch.uint8, device=dev)
    e32.predict_device(d_img.data_ptr(), H, W, d_labels_u8=l32.data_ptr(), stream=st)
    torch.cuda.synchronize()
    got, want = lab.cpu().numpy(), l32.cpu().numpy()
    assert np.array_equal(got, want), "label-exact mode differs from the float32 engine on %d pixels (%r)" % (int((got != want).sum()), stats)
    assert np.array_equal(lab64.cpu().numpy(), want.astype(np.int64))
    # the margin map is the bf16 pass's top-1 minus top-2 logit
    lg = torch.empty((H, W, C), dtype=torch.float32, device=dev)
    eb.predict_device(d_img.data_ptr(), H, W, d_logits=lg.data_ptr(), stream=st)
    torch.cuda.synchronize()
    z = np.sort(lg.cpu().numpy(), -1)
    want_m = z[..., -1] - z[..., -2]
    assert np.array_equal(margin.cpu().numpy(), want_m)
    # a second call (tau calibrated, companion warm) gives the same map
    eb.predict_exact_labels_device(d_img.data_ptr(), H, W, lab.data_ptr(), stream=st)
    torch.cuda.synchronize()
    assert np.array_equal(lab.cpu().numpy(), want)
    if expect_partial is not None:
        assert bool(stats["whole_page_fallback"]) != expect_partial, stats
    eb.close()
    e32.close()
    return stats


@pytest.mark.parametrize("arch,C,shape", [("fcn_skip", 3, (96, 80)), ("fcn_skip", 6, (70, 50)), ("fcn", 3, (160, 96)),
                                          ("fcn_skip", 3, (33, 1)), ("unet", 3, (64, 96)), ("res_unet", 4, (96, 64))])
def test_label_exact_small_pages_every_graph(gpu, oracle_mod, arch, C, shape):
    from pseg_amd import synth
    img = (synth.synth_page(3, shape[0], shape[1], C)[0] if min(shape) >= 32
           else np.random.default_rng(3).integers(0, 256, size=shape, dtype=np.uint8))
    Wt = oracle_mod.init_weights(arch, C, seed=42, gain=1.5, bias_scale=0.05)
    _exact_vs_f32(gpu, arch, C, Wt, img)


def test_label_exact_full_page_random_weights(gpu, oracle_mod):
    """configs[1] page, glorot weights: random-init logits are near-tied nearly everywhere: the cost model
    must hand the whole page to the float32 engine (whole_page_fallback == 1) -- and the result is the float32 map."""
    from pseg_amd import synth
    img = synth.synth_page(0, 2048, 1536, 3)[0]
    Wt = synth.glorot_weights(gpu.Engine("fcn_skip", 3).weight_specs(), seed=42, gain=1.5, bias_scale=0.05)
    stats = _exact_vs_f32(gpu, "fcn_skip", 3, Wt, img, expect_partial=False)
    assert stats["tau"] > 0 and stats["calib_logit_err"] > 0


def _trained_weights(gpu, steps=150):
    from pseg_amd import synth
    e32 = gpu.Engine("fcn_skip", 3, mode=gpu.MODE_F32_EXACT)
    e32.set_weights(synth.glorot_weights(e32.weight_specs(), seed=7))
    e32.train_init(clipnorm=1.0)
    pages = [synth.synth_page(s, 128, 160, 3) for s in range(6)]
    first = last = None
    for it in range(steps):
        img, _, mask = pages[it % len(pages)]
        loss = e32.train_forward_backward(img, mask)[0]
        e32.train_apply(2e-3)
        first = loss if first is None else first
        last = loss
    assert last < 0.7 * first
    Wt = e32.get_weights()
    e32.close()
    return Wt


def _sparse_page(seed, H, W, C=3, frac=0.25):
    """A page with large single-class areas: paper everywhere, the synthetic text / image content only in the top-left
    `frac` x `frac` of the page (title pages, end-of-chapter pages, margins of every scan)."""
    from pseg_amd import synth
    img = synth.synth_page(seed, H, W, C)[0]
    h, w = int(H * frac) // 32 * 32, int(W * frac) // 32 * 32
    rng = np.random.default_rng(seed)
    out = (255 - np.clip(rng.normal(225.0, 8.0, size=(H, W)), 0, 255).astype(np.uint8)).astype(np.uint8)   # synth_page's paper, inverted
    out[:h, :w] = img[:h, :w]
    return out


def test_label_exact_full_page_trained_weights(gpu):
    """150-step-trained weights, a full text page: class boundaries run through nearly every block at the line pitch;
    whichever way the cost model decides, the map equals the float32 engine's and the statistics are consistent."""
    from pseg_amd import synth
    Wt = _trained_weights(gpu)
    img = synth.synth_page(99, 2048, 1536, 3)[0]
    stats = _exact_vs_f32(gpu, "fcn_skip", 3, Wt, img)
    assert 0.0 <= stats["flagged_px_frac"] <= 1.0 and 0.0 <= stats["referee_tile_frac"] <= 1.0
    assert stats["margin_err_running"] >= 0.0 and stats["tau"] >= 2.0 * stats["margin_err_running"] - 1e-6
    print("label-exact, trained weights, text page 2048x1536:", stats)


def _trained_weights_lr(gpu, steps, lr):
    from pseg_amd import synth
    e32 = gpu.Engine("fcn_skip", 3, mode=gpu.MODE_F32_EXACT)
    e32.set_weights(synth.glorot_weights(e32.weight_specs(), seed=7))
    e32.train_init(clipnorm=1.0)
    pages = [synth.synth_page(s, 128, 160, 3) for s in range(6)]
    for it in range(steps):
        img, _, mask = pages[it % len(pages)]
        e32.train_forward_backward(img, mask)
        e32.train_apply(lr)
    Wt = e32.get_weights()
    e32.close()
    return Wt


def test_label_exact_partial_referee_on_a_page_with_large_single_class_areas(gpu):
    """The PARTIAL path at BASELINE.json's page size: trained weights, content in a quarter of the page's width and
    height -- the referee must NOT take the whole page (whole_page_fallback == 0), must re-evaluate well under half of
    it, and the merged map must equal the float32 engine's everywhere (np.array_equal).  ONE training recipe (the bench
    leg's: 200 Adam steps at 1e-3 -- the train step is bit-reproducible, so the weights are the same on every box): a
    regression of the threshold or the cost model cannot hide behind a retry."""
    img = _sparse_page(7, 2048, 1536)
    Wt = _trained_weights_lr(gpu, 200, 1e-3)
    stats = _exact_vs_f32(gpu, "fcn_skip", 3, Wt, img)
    assert stats["whole_page_fallback"] == 0, stats
    assert 0.0 < stats["referee_area_frac"] < 0.5, stats
    assert stats["referee_rects"] >= 1 and stats["referee_cost_vs_full_page"] < 1.0, stats
    assert stats["tau"] >= 2.0 * stats["margin_err_running"] - 1e-6
    print("label-exact, partial referee, 2048x1536:", stats)


def test_label_exact_flag_and_merge_with_a_fixed_threshold_on_a_book_page(gpu, monkeypatch):
    """The flag / cover / merge logic on a realistic layout, independent of the calibration: a book page (text block in the
    middle 55 % x 65 % of the sheet, paper margins around it) with the threshold FIXED through PSEG_EXACT_TAU.  The referee must
    work in parts; every pixel whose bf16 top-2 margin is under the threshold must come out with the float32 engine's label (it
    was flagged, its block refereed, the crop's interior merged), every other pixel with the bf16 OR the float32 label (it lies
    inside a refereed rectangle or not) -- and nothing else.  With this threshold the whole map also equals the float32 one."""
    torch = _torch()
    from pseg_amd import synth
    H, W = 2048, 1536
    rng = np.random.default_rng(5)
    img = (255 - np.clip(rng.normal(225.0, 8.0, size=(H, W)), 0, 255).astype(np.uint8)).astype(np.uint8)
    y0, y1, x0, x1 = 352, 352 + 1344, 352, 352 + 832            # (multiples of 32: the text block's edges are block edges)
    img[y0:y1, x0:x1] = synth.synth_page(31, H, W, 3)[0][y0:y1, x0:x1]
    Wt = _trained_weights_lr(gpu, 200, 1e-3)
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev).cuda_stream
    d_img = torch.from_numpy(img).to(dev)
    e32 = gpu.Engine("fcn_skip", 3, mode=gpu.MODE_F32_EXACT)
    e32.set_weights(Wt)
    l32 = torch.empty((H, W), dtype=torch.uint8, device=dev)
    e32.predict_device(d_img.data_ptr(), H, W, d_labels_u8=l32.data_ptr(), stream=st)
    tau = 1.5
    monkeypatch.setenv("PSEG_EXACT_TAU", repr(tau))
    eb = gpu.Engine("fcn_skip", 3, mode=gpu.MODE_BF16)
    monkeypatch.delenv("PSEG_EXACT_TAU")
    eb.set_weights(Wt)
    lb = torch.empty((H, W), dtype=torch.uint8, device=dev)
    mg = torch.empty((H, W), dtype=torch.float32, device=dev)
    eb.predict_margin_device(d_img.data_ptr(), H, W, mg.data_ptr(), d_labels_u8=lb.data_ptr(), stream=st)
    out = torch.empty((H, W), dtype=torch.uint8, device=dev)
    eb.predict_exact_labels_device(d_img.data_ptr(), H, W, out.data_ptr(), stream=st)
    torch.cuda.synchronize()
    s = eb.label_exact_stats()
    assert s["whole_page_fallback"] == 0 and s["referee_rects"] >= 1 and 0.0 < s["referee_area_frac"] < 0.75, s
    assert abs(s["tau"] - tau) < 1e-6, s
    flagged = mg < tau
    assert 0 < int(flagged.sum()) < H * W
    assert bool((out[flagged] == l32[flagged]).all())                                   # every flagged pixel was refereed and merged
    assert bool(((out == lb) | (out == l32)).all())                                      # nothing but the two engines' labels
    assert bool((out[~flagged & (lb == l32)] == l32[~flagged & (lb == l32)]).all())
    assert torch.equal(out, l32), int((out != l32).sum())                                # 1.5 is far above this net's bf16 error
    eb.close()
    e32.close()


def test_label_exact_threshold_follows_the_running_margin_error(gpu):
    """Every refereed crop is a measurement: after a page has been refereed in parts the threshold is at least twice the
    largest margin change the referee saw, and it stays that high on the next page (until the weights change)."""
    torch = _torch()
    Wt = _trained_weights_lr(gpu, 200, 1e-3)
    dev = torch.device("cuda:0")
    eb = gpu.Engine("fcn_skip", 3, mode=gpu.MODE_BF16)
    eb.set_weights(Wt)
    st = torch.cuda.current_stream(dev).cuda_stream
    seen = []
    for seed in (11, 12, 13):
        img = _sparse_page(seed, 1024, 768, frac=0.4)
        d_img = torch.from_numpy(img).to(dev)
        lab = torch.empty(img.shape, dtype=torch.uint8, device=dev)
        eb.predict_exact_labels_device(d_img.data_ptr(), img.shape[0], img.shape[1], lab.data_ptr(), stream=st)
        torch.cuda.synchronize()
        s = eb.label_exact_stats()
        assert s["tau"] >= 2.0 * s["margin_err_running"] - 1e-6, s
        if s["referee_rects"] > 0:                     # something was refereed in parts: the referee measured
            assert s["margin_err_running"] > 0, s
        seen.append((s["tau"], s["margin_err_running"], s["whole_page_fallback"]))
    assert all(b[1] >= a[1] and b[0] >= a[0] for a, b in zip(seen, seen[1:])), seen   # the running maximum never shrinks between pages
    eb.set_weights(Wt)                                                                  # a weight change resets the evidence
    assert eb.label_exact_stats()["margin_err_running"] == seen[-1][1]                # (until the next call recalibrates)
    img = _sparse_page(14, 1024, 768, frac=0.4)
    d_img = torch.from_numpy(img).to(dev)
    lab = torch.empty(img.shape, dtype=torch.uint8, device=dev)
    eb.predict_exact_labels_device(d_img.data_ptr(), img.shape[0], img.shape[1], lab.data_ptr(), stream=st)
    torch.cuda.synchronize()
    s = eb.label_exact_stats()
    assert s["tau"] >= 2.0 * s["margin_err_running"] - 1e-6
    eb.close()


def test_label_exact_threshold_escalates_when_too_small(gpu, monkeypatch):
    """A threshold far below the bf16 error must be caught by the referee's own check (an unflagged pixel flips inside a
    refereed block) and escalate -- the returned map is still exact."""
    from pseg_amd import synth
    monkeypatch.setenv("PSEG_EXACT_TAU", "1e-7")
    Wt = _trained_weights(gpu, steps=60)
    img = synth.synth_page(5, 512, 384, 3)[0]
    stats = _exact_vs_f32(gpu, "fcn_skip", 3, Wt, img)
    print("escalation:", stats)


def test_network_labels_mode_and_default_is_float32(gpu, oracle_mod):
    """The mirrored API computes in float32 by default, as the reference does (lib/network.py:256-259); exact='labels'
    returns the same `pred` from the throughput engine."""
    from pseg_amd import synth
    from ocr4all_pixel_classifier.lib.network import Network
    from ocr4all_pixel_classifier.lib.dataset import SingleData
    img, binary, _ = synth.synth_page(1, 96, 80, 3)
    Wt = oracle_mod.init_weights("fcn_skip", 3, seed=42, gain=1.5, bias_scale=0.05)
    data = SingleData(image=img, binary=binary, original_shape=img.shape, image_path="p.png")
    lo, po, pr = oracle_mod.predict_single_data("fcn_skip", Wt, img, "f32")
    net = Network("Predict", n_classes=3)
    assert net.model.mode == gpu.MODE_F32_EXACT
    net.model.set_weights(Wt)
    z, p, l = net.predict_single_data(data)
    assert np.array_equal(z, lo) and np.array_equal(l, pr)
    net2 = Network("Predict", n_classes=3, exact="labels")
    net2.model.set_weights(Wt)
    z2, p2, l2 = net2.predict_single_data(data)
    assert np.array_equal(l2, pr) and l2.dtype == np.int64
    assert np.abs(z2 - lo).max() <= 0.02 * max(1.0, float(np.abs(lo).max()))
    assert [np.array_equal(a, pr) for a in net2.predict_labels([img, img])] == [True, True]


def test_label_exact_stream_of_whole_page_referees_skips_the_bf16_pass(gpu, oracle_mod):
    """After three pages in a row whose referee took the whole page, the next eight go to the float32 engine directly (the
    bf16 pass in front of a whole-page referee is wasted); then one page probes, and every probe that falls back again
    doubles the direct stretch (8, 16, 32, ... pages: a stream of text pages pays for the first pass on a vanishing share
    of its pages); every map still equals the float32 engine's; a weight change resets it all."""
    from pseg_amd import synth
    torch = _torch()
    dev = torch.device("cuda:0")
    Wt = oracle_mod.init_weights("fcn_skip", 3, seed=42, gain=1.5, bias_scale=0.05)      # random weights: near-ties everywhere
    eb = gpu.Engine("fcn_skip", 3, mode=gpu.MODE_BF16)
    eb.set_weights(Wt)
    e32 = gpu.Engine("fcn_skip", 3, mode=gpu.MODE_F32_EXACT)
    e32.set_weights(Wt)
    st = torch.cuda.current_stream(dev).cuda_stream
    H, W = 256, 192
    direct = []
    for i in range(31):
        img = synth.synth_page(20 + i, H, W, 3)[0]
        d_img = torch.from_numpy(img).to(dev)
        lab = torch.empty((H, W), dtype=torch.uint8, device=dev)
        l32 = torch.empty((H, W), dtype=torch.uint8, device=dev)
        eb.predict_exact_labels_device(d_img.data_ptr(), H, W, lab.data_ptr(), stream=st)
        e32.predict_device(d_img.data_ptr(), H, W, d_labels_u8=l32.data_ptr(), stream=st)
        torch.cuda.synchronize()
        assert torch.equal(lab, l32), i
        s = eb.label_exact_stats()
        assert s["whole_page_fallback"] == 1
        direct.append(s["direct_float32"])
    assert direct == [0, 0, 0] + [1] * 8 + [0] + [1] * 16 + [0] + [1] * 2, direct
    eb.set_weights(Wt)
    img = synth.synth_page(99, H, W, 3)[0]
    d_img = torch.from_numpy(img).to(dev)
    lab = torch.empty((H, W), dtype=torch.uint8, device=dev)
    eb.predict_exact_labels_device(d_img.data_ptr(), H, W, lab.data_ptr(), stream=st)
    torch.cuda.synchronize()
    assert eb.label_exact_stats()["direct_float32"] == 0
    eb.close()
    e32.close()
