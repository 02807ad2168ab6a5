"""Keras-format HDF5 fixtures for pseg_amd.h5lite (the product's dependency-free reader).

Run with the container's conda interpreter (the only one with h5py):
    /opt/conda/bin/python3.9 tests/golden/make_h5_golden.py
TensorFlow/Keras is not installable offline, so the files are written with h5py the way
tensorflow/python/keras/saving/hdf5_format.py (TF 2.5) does: `layer_names` / `weight_names` attributes
as fixed-length byte-string arrays, one group per layer, datasets named by the variable
("conv2d/kernel:0"), `backend` / `keras_version` attributes; the full-model variant nests them under
/model_weights next to /optimizer_weights and a `model_config` string attribute, as
ModelCheckpoint(save_weights_only=False) does (lib/network.py:177-183).
Writes tests/golden/keras_weights.h5, keras_full_model.h5 and keras_expected.npz (the arrays)."""
import os

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LAYERS = [("input_1", []), ("input_2", []), ("lambda", []),
          ("conv2d_7", [("conv2d_7/kernel:0", (5, 5, 1, 20)), ("conv2d_7/bias:0", (20,))]),
          ("conv2d_8", [("conv2d_8/kernel:0", (3, 3, 20, 6)), ("conv2d_8/bias:0", (6,))]),
          ("max_pooling2d_3", []),
          ("conv2d_transpose_5", [("conv2d_transpose_5/kernel:0", (2, 2, 4, 6)), ("conv2d_transpose_5/bias:0", (4,))]),
          ("concatenate_4", []),
          ("logits", [("logits/kernel:0", (1, 1, 10, 3)), ("logits/bias:0", (3,))])]


def save_weights_to_group(f, rng, expected):
    f.attrs["layer_names"] = [n.encode("utf8") for n, _ in LAYERS]
    f.attrs["backend"] = "tensorflow".encode("utf8")
    f.attrs["keras_version"] = "2.5.0".encode("utf8")
    for lname, ws in LAYERS:
        g = f.create_group(lname)
        g.attrs["weight_names"] = [n.encode("utf8") for n, _ in ws]
        for wname, shape in ws:
            val = rng.standard_normal(shape).astype(np.float32)
            d = g.create_dataset(wname, val.shape, dtype=val.dtype)
            d[:] = val
            expected[wname] = val


def main():
    rng = np.random.default_rng(5)
    exp = {}
    with h5py.File(os.path.join(HERE, "keras_weights.h5"), "w") as f:
        save_weights_to_group(f, rng, exp)
    rng = np.random.default_rng(5)
    with h5py.File(os.path.join(HERE, "keras_full_model.h5"), "w") as f:
        f.attrs["keras_version"] = "2.5.0"                       # str -> variable-length string attribute
        f.attrs["backend"] = "tensorflow"
        f.attrs["model_config"] = '{"class_name": "Functional", "config": {"name": "model"}}'
        save_weights_to_group(f.create_group("model_weights"), rng, {})
        og = f.create_group("optimizer_weights")
        og.attrs["weight_names"] = [b"Adam/iter:0"]
        og.create_dataset("Adam/iter:0", data=np.int64(17))
    np.savez_compressed(os.path.join(HERE, "keras_expected.npz"), **{k.replace("/", "__").replace(":", "--"): v for k, v in exp.items()})
    print("wrote", [os.path.getsize(os.path.join(HERE, n)) for n in ("keras_weights.h5", "keras_full_model.h5", "keras_expected.npz")])


if __name__ == "__main__":
    main()
