"""Host-side mirror of the reference API (no GPU): names, defaults, enum values, record fields,
dataset JSON format, post-processor registry, sharding helper."""
import json
import os

import numpy as np
import pytest

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.json")))


def test_enums_match_reference_values():
    from ocr4all_pixel_classifier.lib.architecture import Architecture, Optimizers, default_preprocess
    assert {m.name: m.value for m in Architecture} == GOLD["architecture_values"]
    assert {m.name: m.value for m in Optimizers} == GOLD["optimizer_values"]
    want = np.array(GOLD["preprocess_f32_bits"], np.uint32).view(np.float32)
    assert np.array_equal(default_preprocess(np.arange(256, dtype=np.uint8)).astype(np.float32), want)
    assert Architecture.FCN_SKIP.model() == "fcn_skip" and Architecture.UNET.preprocess()[1] is False
    with pytest.raises(Exception):
        Architecture.RES_NET.model()            # pretrained backbones are out of scope


def test_util_matches_reference_vectors():
    from ocr4all_pixel_classifier.lib.util import gray_to_rgb, image_to_batch
    img = np.array(GOLD["util_img"], np.uint8)
    assert list(image_to_batch(img).shape) == GOLD["image_to_batch_shape_2d"]
    assert list(image_to_batch(np.zeros((3, 4, 3))).shape) == GOLD["image_to_batch_shape_3d"]
    assert gray_to_rgb(img).tolist() == GOLD["gray_to_rgb"]
    assert list(gray_to_rgb(np.zeros((3, 4, 3), np.uint8)).shape) == GOLD["gray_to_rgb_passthrough_shape"]


def test_records_and_settings_keep_the_reference_fields():
    from ocr4all_pixel_classifier.lib.dataset import SingleData, Dataset
    from ocr4all_pixel_classifier.lib.trainer import TrainSettings, AugmentationSettings
    from ocr4all_pixel_classifier.lib.predictor_data import PredictSettings, Prediction
    from ocr4all_pixel_classifier.lib.metrics import Loss, Monitor
    from ocr4all_pixel_classifier.lib.architecture import Architecture, Optimizers
    from ocr4all_pixel_classifier.lib.colors import ColorMap
    import dataclasses
    # lib/dataset.py:18-29
    assert [f.name for f in dataclasses.fields(SingleData)] == [
        "image", "binary", "orig_binary", "mask", "image_path", "binary_path", "mask_path",
        "line_height_px", "original_shape", "output_path", "user_data"]
    assert SingleData().line_height_px == 1
    ds = Dataset([SingleData(), SingleData()], ColorMap({}))
    assert len(ds) == 2 and len(list(ds)) == 2
    # lib/trainer.py:59-106: positional head + defaults
    assert TrainSettings._fields[:8] == ("n_epoch", "n_classes", "l_rate", "train_data", "validation_data",
                                         "display", "output_dir", "threads")
    d = TrainSettings._field_defaults
    assert d["architecture"] is Architecture.FCN_SKIP and d["loss"] is Loss.CATEGORICAL_CROSSENTROPY
    assert d["monitor"] is Monitor.VAL_LOSS and d["optimizer"] is Optimizers.ADAM
    assert d["optimizer_norm_clipping"] is True and d["optimizer_norm_clip_value"] == 1.0
    assert d["early_stopping_max_performance_drops"] == 10 and d["model_suffix"] == ".h5"
    assert d["data_augmentation"] is False and d["image_dimension"] == 1
    a = AugmentationSettings()
    assert a.rotation_range == 2.5 and a.zoom_range == [0.95, 1.05]
    assert a.to_image_params()["interpolation_order"] == 3 and "brightness_range" in a.to_image_params()
    assert a.to_mask_params()["interpolation_order"] == 0 and "brightness_range" not in a.to_mask_params()
    # lib/predictor_data.py:18-26
    ps = PredictSettings()
    assert ps.n_classes == -1 and ps.high_res_output is False and ps.post_process is None
    assert Prediction._fields == ("labels", "probabilities", "data")
    assert Monitor.JACRAD_COEF.value == "jacard_coef" and Loss.CATEGORCAL_FOCAL.value == "categorical_focal"


def test_postprocessor_registry_key_normalisation():
    from ocr4all_pixel_classifier.lib import postprocess as pp
    assert pp.find_postprocessor("cc_majority") is pp.vote_connected_component_class
    assert pp.find_postprocessor("CC-Vote") is pp.vote_connected_component_class
    assert pp.find_postprocessor("bounding_boxes") is pp.add_bounding_boxes
    assert pp.find_postprocessor("BBox") is pp.add_bounding_boxes
    with pytest.raises(KeyError):
        pp.find_postprocessor("nope")
    assert "cc_majority" in pp.postprocess_help()


def test_dataset_json_and_prepare_images(tmp_path):
    from PIL import Image
    from ocr4all_pixel_classifier.lib.dataset import DatasetLoader, prepare_images
    from ocr4all_pixel_classifier.lib.colors import ColorMap
    rng = np.random.default_rng(0)
    gray = rng.integers(0, 256, (40, 30), dtype=np.uint8)
    mask_rgb = np.zeros((40, 30, 3), np.uint8)
    mask_rgb[5:20, 3:12] = (255, 0, 0)
    Image.fromarray(gray).save(tmp_path / "p.png")
    Image.fromarray(mask_rgb).save(tmp_path / "m.png")
    cm = ColorMap({"(0, 0, 0)": [0, "bg"], "(255, 0, 0)": [1, "text"]})
    js = {"train": [{"binary_path": str(tmp_path / "p.png"), "image_path": str(tmp_path / "p.png"),
                     "mask_path": str(tmp_path / "m.png"), "line_height_px": 6}], "test": [], "eval": []}
    (tmp_path / "d.json").write_text(json.dumps(js))            # README.md:46-70 format
    import pseg_amd
    from pseg_amd import engine as E
    # geometry helpers are host arithmetic and need no GPU
    assert E.rescale_shape((61, 83), 6 / 23) == (16, 22) and E.rescale_shape((5, 7), 0.5) == (2, 4)   # half to even
    (wy, ry), (wx, rx) = E.aa_kernels((61, 83), (16, 22))
    assert ry == int(4 * ((61 / 16 - 1) / 2) + 0.5) and len(wy) == 2 * ry + 1 and abs(wy.sum() - 1) < 1e-12
    assert E.aa_kernels((10, 10), (20, 10)) == [(None, 0), (None, 0)]
    if pseg_amd.device_count() == 0:
        # the pixel work of the loader (nearest / Gaussian / bicubic) lives on the GPU: loud error, no CPU fallback
        with pytest.raises(Exception, match="GPU|HIP"):
            DatasetLoader(6, cm).load_data_from_json([str(tmp_path / "d.json")], "train")
        with pytest.raises(Exception, match="GPU|HIP"):
            prepare_images(gray, np.where(gray > 127, 255, 0).astype(np.uint8), 12, 6)
        return
    ds = DatasetLoader(6, cm).load_data_from_json([str(tmp_path / "d.json")], "train")
    assert len(ds) == 1 and ds.data[0].image.shape == (40, 30)


def test_network_needs_the_gpu_engine():
    import pseg_amd
    from ocr4all_pixel_classifier.lib.network import Network
    if pseg_amd.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(Exception):
        Network("Predict", n_classes=3)


def test_shard_pages():
    from pseg_amd.parallel import shard_pages
    assert shard_pages(10, 0, 4) == [0, 4, 8] and shard_pages(10, 3, 4) == [3, 7]
    assert shard_pages(2, 3, 4) == [] and shard_pages(0, 0, 1) == []
    allp = sorted(sum((shard_pages(257, r, 8) for r in range(8)), []))
    assert allp == list(range(257))
    with pytest.raises(ValueError):
        shard_pages(4, 4, 4)


def test_synthetic_weights_agree_with_oracle_init(oracle_mod):
    """pseg_amd.synth.glorot_weights (product side, used by bench / Network) draws the same
    numbers as the oracle's init given the same spec order."""
    from pseg_amd import synth
    for arch in ("fcn_skip", "res_unet"):
        specs = []
        for name, kind, shp, cout in oracle_mod.weight_specs(arch, 3):
            specs += [(name + "/kernel", tuple(shp)), (name + "/bias", (cout,))]
        a = synth.glorot_weights(specs, seed=42, gain=1.5, bias_scale=0.05)
        b = oracle_mod.init_weights(arch, 3, seed=42, gain=1.5, bias_scale=0.05)
        assert list(a.keys()) == list(b.keys())
        for k in a:
            assert np.array_equal(a[k], b[k]), k
    img, binary, mask = synth.synth_page(3, 128, 96, 6)
    assert img.shape == (128, 96) and img.dtype == np.uint8 and set(np.unique(binary)) <= {0, 1}
    assert mask.max() <= 5 and np.array_equal(synth.synth_page(3, 128, 96, 6)[0], img)


# ---- round 2: model files decide the graph; Keras layer order; lazy probabilities (no GPU needed) ----------------------------
class _FakeEngine:
    def __init__(self, specs):
        self._specs, self.got = specs, None

    def weight_specs(self):
        return self._specs

    def set_weights(self, w):
        self.got = w


def _res_unet_specs(C=3):
    """res_unet weight table in the engine's (Keras creation) order: stem (k3, k3, k1), 8 residual blocks (conv_a, conv_b,
    shortcut), 2 bridge convs, logits."""
    f = [32, 64, 128, 256, 512]
    specs, n = [], [0]

    def conv(cin, cout, k=3):
        name = "conv2d" if n[0] == 0 else "conv2d_%d" % n[0]
        n[0] += 1
        specs.append((name + "/kernel", (k, k, cin, cout)))
        specs.append((name + "/bias", (cout,)))
    conv(1, f[0]); conv(f[0], f[0]); conv(1, f[0], 1)
    cin = f[0]
    for l in range(1, 5):
        conv(cin, f[l]); conv(f[l], f[l]); conv(cin, f[l]); cin = f[l]
    conv(f[4], f[4]); conv(f[4], f[4])
    for l, skip in ((4, f[3]), (3, f[2]), (2, f[1]), (1, f[0])):
        c2 = cin + skip
        conv(c2, f[l]); conv(f[l], f[l]); conv(c2, f[l]); cin = f[l]
    specs.append(("logits/kernel", (1, 1, f[1], C)))
    specs.append(("logits/bias", (C,)))
    return specs


def test_network_reads_graph_and_classes_from_the_model_file(tmp_path):
    from ocr4all_pixel_classifier.lib.network import Network
    rng = np.random.default_rng(0)
    mk = lambda sh: rng.standard_normal(sh).astype(np.float32)
    skip = [("conv2d/kernel:0", mk((5, 5, 1, 20))), ("conv2d_transpose/kernel:0", mk((5, 5, 80, 80))),
            ("conv2d_transpose_2/kernel:0", mk((5, 5, 40, 120))), ("logits/kernel:0", mk((1, 1, 50, 6)))]
    assert Network._graph_of_file("m.npz", skip) == ("fcn_skip", 6, 1)
    plain = [("conv2d/kernel:0", mk((5, 5, 3, 20))), ("conv2d_transpose/kernel:0", mk((5, 5, 80, 80))),
             ("conv2d_transpose_2/kernel:0", mk((5, 5, 40, 60))), ("logits/kernel:0", mk((1, 1, 20, 3)))]
    assert Network._graph_of_file("m.npz", plain) == ("fcn", 3, 3)
    assert Network._graph_of_file("m.npz", [("conv2d/kernel:0", mk((3, 3, 1, 64))), ("logits/kernel:0", mk((1, 1, 64, 4)))]) == ("unet", 4, 1)
    assert Network._graph_of_file("m.npz", [("conv2d/kernel:0", mk((3, 3, 1, 32))), ("logits/kernel:0", mk((1, 1, 64, 2)))]) == ("res_unet", 2, 1)
    # a full-model Keras file names its model: the golden fixture's is Keras' default 'model' -> the shapes decide
    from pseg_amd import h5lite
    full = os.path.join(os.path.dirname(__file__), "golden", "keras_full_model.h5")
    assert h5lite.read_keras_model_name(full) == "model"
    assert h5lite.read_keras_model_name(os.path.join(os.path.dirname(__file__), "golden", "keras_weights.h5")) is None


def test_res_unet_files_are_matched_in_keras_layer_order(tmp_path):
    """Keras lists a residual block as (conv_a, shortcut, conv_b) -- depth order, ties by traversal from Add([shortcut, res]) --
    while the engine's table is creation order (conv_a, conv_b, shortcut): loading goes by the numeric name suffix per layer
    class, whatever offset the per-process counters carry; saving emits Keras' order."""
    from ocr4all_pixel_classifier.lib.network import Network
    from pseg_amd import h5lite
    specs = _res_unet_specs()
    rng = np.random.default_rng(1)
    want = {n: rng.standard_normal(sh).astype(np.float32) * 0.01 for n, sh in specs}
    net = Network.__new__(Network)
    net.architecture = "res_unet"
    net.model = _FakeEngine(specs)
    layers = []
    for n, _ in specs:
        ln = n.split("/")[0]
        if not layers or layers[-1] != ln:
            layers.append(ln)
    keras = net._keras_layer_order(layers)
    assert keras[:6] == ["conv2d", "conv2d_1", "conv2d_2", "conv2d_3", "conv2d_5", "conv2d_4"] and keras[-1] == "logits"
    assert sorted(keras) == sorted(layers)
    # a "trained" file: Keras order, counters offset by 14
    def shifted(ln):
        if ln == "logits":
            return ln
        base, _, suf = ln.rpartition("_")
        k = int(suf) if suf.isdigit() else 0
        return "conv2d_%d" % (k + 14)
    file_layers = [("input_1", [])] + [(shifted(ln), [(shifted(ln) + "/kernel:0", want[ln + "/kernel"]), (shifted(ln) + "/bias:0", want[ln + "/bias"])])
                                       for ln in keras]
    path = str(tmp_path / "res_unet.h5")
    h5lite.write_keras_weights(path, file_layers)
    file_w = Network._read_weight_file(path)
    assert Network._graph_of_file(path, file_w)[0] == "res_unet"
    net._set_file_weights(path, file_w)
    assert set(net.model.got) == set(want) and all(np.array_equal(net.model.got[k], want[k]) for k in want)
    # positional matching (round 1) would have put the shortcut kernel where conv_b's belongs: shapes differ there
    assert want["conv2d_4/kernel"].shape != want["conv2d_5/kernel"].shape


def test_prediction_probabilities_can_be_lazy():
    from ocr4all_pixel_classifier.lib.predictor_data import LazyArray, Prediction
    calls = []
    p = Prediction(np.zeros((2, 2), np.int64), LazyArray(lambda: calls.append(1) or np.ones((2, 2, 3), np.float32)), "data")
    assert calls == []
    labels, prob, data = p
    assert prob.shape == (2, 2, 3) and p.probabilities is prob and p[1] is prob and calls == [1] and data == "data"
    q = Prediction(labels, prob, data)
    assert q.probabilities is prob and len(q) == 3 and q._fields == ("labels", "probabilities", "data")


def test_batch_units_ramp_at_head_and_tail():
    """pseg_batch_units (the partition pseg_predict_batch uses; lib/predictor.py:27-30 is a page-by-page loop): units cover the list
    in order, never mix shapes, never exceed the cap; a same-shape run that opens the list starts 1, 2, 4 ..., one that closes it
    ends ... 4, 2, 1; interior runs go in units of `cap`."""
    from pseg_amd import engine as E
    A, B = (2048, 1536), (1000, 700)
    assert E.batch_units([], 8) == []
    assert E.batch_units([A], 8) == [(0, 1)]
    u = E.batch_units([A] * 32, 8)
    assert [c for _, c in u] == [1, 2, 4, 8, 8, 2, 4, 2, 1] and [f for f, _ in u] == [0, 1, 3, 7, 15, 23, 25, 29, 31]
    assert [c for _, c in E.batch_units([A] * 8, 2)] == [1, 2, 2, 2, 1]
    assert [c for _, c in E.batch_units([A] * 5, 1)] == [1] * 5
    mixed = [A] * 10 + [B] * 3 + [A] * 20 + [B] * 11
    u = E.batch_units(mixed, 8)
    assert sum(c for _, c in u) == len(mixed) and all(f == sum(c for _, c in u[:i]) for i, (f, _) in enumerate(u))
    for f, c in u:
        assert 1 <= c <= 8 and len({mixed[i] for i in range(f, f + c)}) == 1
    heads = [c for f, c in u if f < 10]
    assert heads[:3] == [1, 2, 4]                               # the opening run ramps up ...
    assert [c for f, c in u if 13 <= f < 33] == [8, 8, 4]       # ... an interior run does not
    assert [c for f, c in u if f >= 33][-3:] == [4, 2, 1]       # ... the closing run ramps down
    with pytest.raises(E.PsegError):
        E.batch_units([A], 0)


def test_augmentation_settings_reach_the_generators_as_in_the_reference():
    """lib/trainer.py:14-56: to_image_params carries brightness_range and interpolation order 3, to_binary_params / to_mask_params
    order 0 WITHOUT the brightness key; every fill mode keras-preprocessing accepts is taken, anything else raises; the brightness
    draw is the last of get_random_transform (the affine parameters under a shared seed stay equal across the three generators)."""
    from ocr4all_pixel_classifier.lib.trainer import AugmentationSettings
    from ocr4all_pixel_classifier.lib.data_generator import ImageDataGeneratorCustom as G
    s = AugmentationSettings(brightness_range=[0.7, 1.3], image_fill_mode='reflect', binary_fill_mode='wrap', mask_fill_mode='constant', mask_cval=9)
    pi, pb, pm = s.to_image_params(), s.to_binary_params(), s.to_mask_params()
    assert pi['brightness_range'] == [0.7, 1.3] and pi['interpolation_order'] == 3 and pi['fill_mode'] == 'reflect'
    assert 'brightness_range' not in pb and pb['interpolation_order'] == 0 and pb['fill_mode'] == 'wrap'
    assert 'brightness_range' not in pm and pm['fill_mode'] == 'constant' and pm['cval'] == 9
    gi, gb, gm = (G(**p, data_format='channels_last') for p in (pi, pb, pm))
    ti, tb, tm = (g.get_random_transform((100, 140, 1), seed=77) for g in (gi, gb, gm))
    assert tb['brightness'] is None and tm['brightness'] is None and 0.7 <= ti['brightness'] <= 1.3
    assert all(ti[k] == tb[k] == tm[k] for k in ('theta', 'tx', 'ty', 'shear', 'zx', 'zy', 'flip_horizontal', 'flip_vertical'))
    for bad in ('mirror', 'grid-wrap', ''):
        with pytest.raises(Exception):
            G(fill_mode=bad)
    with pytest.raises(ValueError):
        G(brightness_range=1.2)
    assert AugmentationSettings().brightness_range is None          # the reference default: no brightness draw at all
