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
    """configs[1] page, glorot weights: random-init logits are near-tied nearly everywhere, the referee may
    take the whole page -- the result must still be the float32 map."""
    from pseg_amd import synth
    img = synth.synth_page(0, 2048, 1536, 3)[0]
    Wt = synth.glorot_weights(gpu.Engine("fcn_skip", 3).weight_specs(), seed=42, gain=1.5, bias_scale=0.05)
    stats = _exact_vs_f32(gpu, "fcn_skip", 3, Wt, img)
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


def test_label_exact_full_page_trained_weights(gpu):
    """150-step-trained weights: confident regions keep their bf16 labels, only the blocks along class boundaries go
    through the referee; the map equals the float32 engine's and the statistics are consistent."""
    from pseg_amd import synth
    Wt = _trained_weights(gpu)
    img = synth.synth_page(99, 2048, 1536, 3)[0]
    stats = _exact_vs_f32(gpu, "fcn_skip", 3, Wt, img)
    assert 0.0 <= stats["flagged_px_frac"] <= 1.0 and 0.0 <= stats["referee_tile_frac"] <= 1.0
    print("label-exact, trained weights, 2048x1536:", stats)


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
