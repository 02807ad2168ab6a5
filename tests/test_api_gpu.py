"""The drop-in Python API (ocr4all_pixel_classifier.lib.*) end to end on the GPU, plus full-size
(BASELINE.json sizes) property checks where the oracle is too slow to run."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(gpu, oracle_mod, exact, n_classes=3, shape=(96, 80), page=0):
    from pseg_amd import synth
    from ocr4all_pixel_classifier.lib.network import Network
    from ocr4all_pixel_classifier.lib.dataset import SingleData
    img, binary, mask = synth.synth_page(page, shape[0], shape[1], n_classes)
    net = Network("Predict", n_classes=n_classes, exact=exact)
    Wt = oracle_mod.init_weights("fcn_skip", n_classes, seed=42, gain=1.5, bias_scale=0.05)
    net.model.set_weights(Wt)
    data = SingleData(image=img, binary=binary, original_shape=img.shape, image_path="page.png")
    return net, Wt, data


def test_predictor_roundtrip_exact_mode(gpu, oracle_mod, tmp_path):
    from ocr4all_pixel_classifier.lib.predictor import Predictor
    from ocr4all_pixel_classifier.lib.predictor_data import PredictSettings
    from ocr4all_pixel_classifier.lib.postprocess import find_postprocessor
    from ocr4all_pixel_classifier.lib.colors import ColorMap
    from ocr4all_pixel_classifier.lib.dataset import Dataset
    net, Wt, data = _setup(gpu, oracle_mod, exact=True)
    cm = ColorMap({"(255, 255, 255)": [0, "bg"], "(255, 0, 0)": [1, "text"], "(0, 255, 0)": [2, "image"]})
    settings = PredictSettings(n_classes=3, color_map=cm, post_process=[find_postprocessor("cc_majority")],
                               output=str(tmp_path))
    pred = Predictor(settings, net)
    p = pred.predict_single(data)
    logit_o, prob_o, lab_o = oracle_mod.predict_single_data("fcn_skip", Wt, data.image, "f32")
    want = oracle_mod.vote_connected_component_class(lab_o, data.binary)
    assert p.labels.dtype == np.int64 and np.array_equal(p.labels, want)
    assert np.abs(p.probabilities - prob_o).max() <= 2e-6
    assert [q.labels.shape for q in pred.predict(Dataset([data], cm))] == [data.image.shape]
    m = pred.predict_masks(data)
    wc, wo, wi, wf = oracle_mod.generate_output_masks(want, data.binary, cm.lut())
    assert np.array_equal(m.color, wc) and np.array_equal(m.overlay, wo)
    assert np.array_equal(m.inverted_overlay, wi) and np.array_equal(m.fg_color_mask, wf)
    for sub in ("color", "overlay", "inverted"):
        assert (tmp_path / sub).is_dir()
    from ocr4all_pixel_classifier.lib.output import output_data
    output_data(str(tmp_path), want, data, cm)
    assert (tmp_path / "color" / "page.png").exists()
    # weights round-trip through the engine and a Keras HDF5 file (lib/network.py:59: no extension = .h5)
    path = net.save_weights(str(tmp_path / "model.h5"))
    assert path.endswith(".h5") and open(path, "rb").read(8) == b"\x89HDF\r\n\x1a\n"
    from ocr4all_pixel_classifier.lib.network import Network
    net2 = Network("Predict", n_classes=3, model=str(tmp_path / "model"), exact=True)
    got = net2.model.get_weights()
    assert all(np.array_equal(got[k], Wt[k]) for k in Wt)
    _, _, lab2 = net2.predict_single_data(data)
    assert np.array_equal(lab2, lab_o)
    # a trained file carries Keras' per-process layer counters (conv2d_14/...): matched by order, like load_weights
    from pseg_amd import h5lite
    layers = h5lite.read_keras_weights(path)
    renamed = [("x%d_%s" % (i, n), [("x%d_%s" % (i, w), a) for w, a in ws]) for i, (n, ws) in enumerate(layers)]
    h5lite.write_keras_weights(str(tmp_path / "renamed.h5"), [("input_1", [])] + renamed)
    net3 = Network("Predict", n_classes=3, model=str(tmp_path / "renamed.h5"), exact=True)
    assert all(np.array_equal(net3.model.get_weights()[k], Wt[k]) for k in Wt)
    with pytest.raises(Exception, match="weight tensors|shape"):
        Network("Predict", n_classes=6, model=str(tmp_path / "model.h5"), exact=True)     # wrong class count
    # the native .npz format still works
    assert net.save_weights(str(tmp_path / "m2.npz")).endswith(".npz")
    net4 = Network("Predict", n_classes=3, model=str(tmp_path / "m2.npz"), exact=True)
    assert all(np.array_equal(net4.model.get_weights()[k], Wt[k]) for k in Wt)


def test_char_height_via_api(gpu, oracle_mod, tmp_path):
    from PIL import Image
    from pseg_amd import synth
    from ocr4all_pixel_classifier.lib.image_ops import compute_char_height
    img, _, _ = synth.synth_page(1, 384, 512)
    gray = 255 - img
    Image.fromarray(gray).save(tmp_path / "scan.png")
    assert compute_char_height(str(tmp_path / "scan.png"), False) == oracle_mod.compute_char_height_from_gray(gray, False)
    with pytest.raises(Exception):
        compute_char_height(str(tmp_path / "missing.png"), False)


def test_full_size_page_properties(gpu, oracle_mod):
    """BASELINE configs[1] size (2048x1536): the oracle would take too long for every check, so
    use size-independent properties: determinism, exact-vs-bf16 agreement outside near-ties,
    tiling invariance (an interior crop far from the borders sees the same receptive field),
    CC vote idempotence and label-set preservation, masks consistency."""
    from pseg_amd import synth
    H, W, C = 2048, 1536, 3
    img, binary, _ = synth.synth_page(0, H, W, C)
    Wt = oracle_mod.init_weights("fcn_skip", C, seed=42, gain=1.5, bias_scale=0.05)
    e32 = gpu.Engine("fcn_skip", C, mode=gpu.MODE_F32_EXACT)
    e32.set_weights(Wt)
    eb = gpu.Engine("fcn_skip", C, mode=gpu.MODE_BF16)
    eb.set_weights(Wt)
    z32, _, l32 = e32.predict(img, want_probs=False)
    zb, _, lb = eb.predict(img, want_probs=False)
    zb2, _, lb2 = eb.predict(img, want_probs=False)
    assert np.array_equal(zb, zb2) and np.array_equal(lb, lb2)            # deterministic
    err = np.abs(zb - z32).max()
    assert err <= 0.03 * max(1.0, np.abs(z32).max())                      # bf16 vs f32, whole net
    srt = np.sort(z32, -1)
    margin = srt[..., -1] - srt[..., -2]
    assert not ((lb != l32) & (margin > 2 * err)).any()                   # flips only at near-ties
    assert (lb != l32).mean() < 0.02
    # the oracle agrees with the exact engine on a crop-sized page (bit-identical), and an
    # interior window of the full page equals the same window of a 512x512 sub-page whose borders
    # are > 80 px (receptive-field radius 72) away
    y0, x0 = 768, 512
    sub = np.ascontiguousarray(img[y0:y0 + 512, x0:x0 + 512])
    zs, _, _ = e32.predict(sub, want_probs=False)
    assert np.array_equal(zs[96:416, 96:416], z32[y0 + 96:y0 + 416, x0 + 96:x0 + 416])
    zo = oracle_mod.forward("fcn_skip", Wt, sub[:160, :192])
    zg, _, _ = e32.predict(np.ascontiguousarray(sub[:160, :192]), want_probs=False)
    assert np.array_equal(zo, zg)
    # post-process properties at full size
    v1 = gpu.cc_vote(lb.copy(), binary, C)
    v2 = gpu.cc_vote(v1.copy(), binary, C)
    assert np.array_equal(v1, v2)                                          # idempotent
    assert np.array_equal(v1[binary == 0], lb[binary == 0])                # paper pixels untouched
    assert set(np.unique(v1)) <= set(np.unique(lb))
    lut = np.array([[255, 255, 255], [255, 0, 0], [0, 255, 0]], np.uint8)
    color, overlay, inverted, fg = gpu.masks(v1, binary, lut)
    assert np.array_equal(color, lut[v1])
    assert (overlay[binary == 1] == 0).all() and np.array_equal(inverted, fg)
    assert int(overlay.astype(np.int64).sum() + inverted.astype(np.int64).sum()) == int(color.astype(np.int64).sum())
    e32.close()
    eb.close()


def test_config5_size_pipeline_properties(gpu, oracle_mod):
    """BASELINE configs[4] size: 4096x3072, 6 classes, predict + cc_majority vote + masks on the GPU.  The
    oracle cannot run this size in test time, so: determinism, interior-window agreement with a sub-page
    (the receptive field is local), vote idempotence / label-set preservation, mask identities, and the
    uint8 label map equal to the int64 one."""
    import torch
    from pseg_amd import synth
    H, W, C = 4096, 3072, 6
    img, binary, _ = synth.synth_page(5, H, W, C)
    Wt = oracle_mod.init_weights("fcn_skip", C, seed=42, gain=1.5, bias_scale=0.05)
    eb = gpu.Engine("fcn_skip", C, mode=gpu.MODE_BF16)
    eb.set_weights(Wt)
    _, _, lab = eb.predict(img, want_logits=False, want_probs=False)
    _, _, lab2 = eb.predict(img, want_logits=False, want_probs=False)
    assert lab.shape == (H, W) and lab.dtype == np.int64 and np.array_equal(lab, lab2)
    assert 0 <= lab.min() and lab.max() < C
    # device entry with a uint8 label map
    d_img = torch.from_numpy(img).cuda()
    d_u8 = torch.empty((H, W), dtype=torch.uint8, device="cuda")
    eb.predict_device(d_img.data_ptr(), H, W, d_labels_u8=d_u8.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(d_u8.cpu().numpy(), lab)
    # interior window == same window of a 640x640 sub-page (borders > 72 px away)
    y0, x0 = 2048, 1536
    sub = np.ascontiguousarray(img[y0:y0 + 640, x0:x0 + 640])
    zs, _, ls = eb.predict(sub, want_probs=False)
    zf, _, _ = eb.predict(np.ascontiguousarray(img[y0 - 256:y0 + 896, x0 - 256:x0 + 896]), want_probs=False)
    assert np.array_equal(zs[96:544, 96:544], zf[256 + 96:256 + 544, 256 + 96:256 + 544])
    assert np.array_equal(ls[96:544, 96:544], lab[y0 + 96:y0 + 544, x0 + 96:x0 + 544])
    # post-process at full size
    v1 = gpu.cc_vote(lab.copy(), binary, C)
    assert np.array_equal(gpu.cc_vote(v1.copy(), binary, C), v1)
    assert np.array_equal(v1[binary == 0], lab[binary == 0]) and set(np.unique(v1)) <= set(np.unique(lab))
    lut = np.array([[255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 0], [0, 255, 255]], np.uint8)
    color, overlay, inverted, fg = gpu.masks(v1, binary, lut)
    assert np.array_equal(color, lut[v1]) and (overlay[binary == 1] == 0).all() and np.array_equal(inverted, fg)
    eb.close()


# ---- the Predictor chain on the device (pseg_predict_chain) against the stage-by-stage host chain -----------------------------

def _host_chain(pred, data, want_masks):
    """The reference's chain stage by stage through host arrays (lib/predictor.py:32-54), as rounds 1-2 ran it."""
    from ocr4all_pixel_classifier.lib.output import generate_output_masks
    d, _, lab = pred._labels(data)
    return (d, lab, generate_output_masks(d, lab, pred.settings.color_map) if want_masks else None)


@pytest.mark.parametrize("exact", [True, "labels", False])
@pytest.mark.parametrize("posts,high_res", [(["cc_majority"], False), (["bounding_boxes"], False), (["cc_majority", "bounding_boxes"], True),
                                            ([], True), (["bounding_boxes", "cc_majority"], False)])
def test_predictor_device_chain_equals_the_host_chain(gpu, oracle_mod, exact, posts, high_res):
    """Predictor.predict_single / predict_masks keep the uint8 label map on the device from the network through
    scale_to_original_shape, the registered post-processors and generate_output_masks; the results must be array_equal to
    the stage-by-stage host chain (int64 NumPy maps between the stages, as the reference has them)."""
    from ocr4all_pixel_classifier.lib.predictor import Predictor
    from ocr4all_pixel_classifier.lib.predictor_data import PredictSettings
    from ocr4all_pixel_classifier.lib.postprocess import find_postprocessor
    from ocr4all_pixel_classifier.lib.colors import ColorMap
    import dataclasses
    net, Wt, data = _setup(gpu, oracle_mod, exact=exact, shape=(160, 96), page=4)
    if high_res:
        rng = np.random.default_rng(1)
        orig = (231, 137)                                   # the scan's own resolution: not a multiple of anything
        data = dataclasses.replace(data, original_shape=orig, orig_binary=(rng.random(orig) < 0.2).astype(np.uint8))
    cm = ColorMap({"(255, 255, 255)": [0, "bg"], "(255, 0, 0)": [1, "text"], "(0, 255, 0)": [2, "image"]})
    settings = PredictSettings(n_classes=3, color_map=cm, post_process=[find_postprocessor(p) for p in posts], high_res_output=high_res)
    pred = Predictor(settings, net)
    assert pred._chain_ops() == [{"cc_majority": "cc_vote", "bounding_boxes": "bbox"}[p] for p in posts]
    d_h, lab_h, masks_h = _host_chain(pred, data, True)
    p = pred.predict_single(data)
    assert p.labels.dtype == np.int64 and p.labels.shape == lab_h.shape and np.array_equal(p.labels, lab_h)
    assert np.array_equal(np.asarray(p.data.binary), np.asarray(d_h.binary)) and np.array_equal(p.data.image, d_h.image)
    lab2, prob2, d2 = p                                         # unpacking resolves the lazy fields
    assert np.array_equal(lab2, lab_h) and prob2.shape == data.image.shape + (3,)
    m = pred.predict_masks(data)
    for got, want in zip((m.color, m.overlay, m.inverted_overlay, m.fg_color_mask),
                         (masks_h.color, masks_h.overlay, masks_h.inverted_overlay, masks_h.fg_color_mask)):
        assert got.dtype == np.uint8 and np.array_equal(got, want)
    # results stay valid after later calls (the recycled pinned blocks are not handed out twice while alive)
    keep = p.labels.copy()
    for _ in range(3):
        pred.predict_masks(data)
        pred.predict_single(data)
    assert np.array_equal(p.labels, keep) and np.array_equal(m.color, masks_h.color)


def test_predictor_chain_falls_back_for_foreign_postprocessors(gpu, oracle_mod):
    from ocr4all_pixel_classifier.lib.predictor import Predictor
    from ocr4all_pixel_classifier.lib.predictor_data import PredictSettings
    from ocr4all_pixel_classifier.lib.postprocess import find_postprocessor
    from ocr4all_pixel_classifier.lib.colors import ColorMap
    net, Wt, data = _setup(gpu, oracle_mod, exact=True)
    seen = []

    def mine(pred, d):
        seen.append(pred.dtype)
        return np.where(pred == 1, 2, pred)

    cm = ColorMap({"(255, 255, 255)": [0, "bg"], "(255, 0, 0)": [1, "text"], "(0, 255, 0)": [2, "image"]})
    pred = Predictor(PredictSettings(n_classes=3, color_map=cm, post_process=[find_postprocessor("cc_majority"), mine]), net)
    assert pred._chain_ops() is None
    p = pred.predict_single(data)
    assert seen == [np.dtype(np.int64)] and not (p.labels == 1).any()
    lab_o = oracle_mod.predict_single_data("fcn_skip", Wt, data.image, "f32")[2]
    want = oracle_mod.vote_connected_component_class(lab_o, data.binary)
    assert np.array_equal(p.labels, np.where(want == 1, 2, want))


def test_engine_trim_and_status(gpu, oracle_mod):
    """pseg_engine_trim frees every page slot of the activation tensors (an engine otherwise keeps the largest canvas x slot count it
    has seen) and the next call allocates what its page needs -- same label maps before and after, for a single page and for a unit;
    pseg_engine_status on a healthy engine is PSEG_OK and a NULL engine an error."""
    import ctypes
    from pseg_amd import synth
    from pseg_amd import engine as E
    eng = gpu.Engine("fcn_skip", 3, mode=gpu.MODE_BF16)
    eng.set_weights(synth.glorot_weights(eng.weight_specs(), seed=42, gain=1.5, bias_scale=0.05))
    pages = [synth.synth_page(30 + i, 160, 224, 3)[0] for i in range(6)]
    one = [eng.predict(p, want_logits=False, want_probs=False)[2] for p in pages]
    unit = eng.predict_batch(pages, dtype=np.uint8)                      # grows the tensors to several slots
    assert all(np.array_equal(u, o) for u, o in zip(unit, one))
    eng.status()
    eng.trim()
    with pytest.raises(gpu.PsegError):
        eng.activation("conv2d_2")                                       # no canvas: nothing to read back
    again = eng.predict(pages[3], want_logits=False, want_probs=False)[2]
    assert np.array_equal(again, one[3])
    unit2 = eng.predict_batch(pages, dtype=np.uint8)
    assert all(np.array_equal(u, o) for u, o in zip(unit2, one))
    eng.trim(); eng.trim()                                               # idempotent
    eng.status()
    assert E.lib().pseg_engine_status(None, None) != 0 and E.lib().pseg_engine_trim(None) != 0
    eng.close()
