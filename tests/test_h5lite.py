"""pseg_amd.h5lite: dependency-free reader / writer for Keras HDF5 weight files, against fixtures
written by h5py the way Keras does (tests/golden/make_h5_golden.py) and against h5py itself when the
container's conda interpreter is present."""
import os
import subprocess

import numpy as np
import pytest

from pseg_amd import h5lite

G = os.path.join(os.path.dirname(__file__), "golden")
EXP = {k.replace("__", "/").replace("--", ":"): v for k, v in
       np.load(os.path.join(G, "keras_expected.npz"), allow_pickle=False).items()}
ORDER = ["input_1", "input_2", "lambda", "conv2d_7", "conv2d_8", "max_pooling2d_3", "conv2d_transpose_5", "concatenate_4", "logits"]


@pytest.mark.parametrize("fn", ["keras_weights.h5", "keras_full_model.h5"])
def test_reads_keras_files(fn):
    layers = h5lite.read_keras_weights(os.path.join(G, fn))
    assert [n for n, _ in layers] == ORDER                       # file order = model.layers order
    seen = 0
    for lname, ws in layers:
        for wname, arr in ws:
            assert wname.startswith(lname + "/") and arr.dtype == np.float32
            assert np.array_equal(arr, EXP[wname])
            seen += 1
    assert seen == len(EXP) == 8
    assert [len(ws) for _, ws in layers] == [0, 0, 0, 2, 2, 0, 2, 0, 2]


def test_full_model_file_attributes_and_other_groups():
    f = h5lite.H5File(os.path.join(G, "keras_full_model.h5"))
    root = f.obj(f.root)
    assert set(root.links) == {"model_weights", "optimizer_weights"}
    assert bytes(root.attrs["backend"]) == b"tensorflow" and b"Functional" in bytes(root.attrs["model_config"])
    og = f.obj(f.group("/optimizer_weights"))
    assert [bytes(x) for x in og.attrs["weight_names"]] == [b"Adam/iter:0"]
    assert int(f.dataset(f.group("/optimizer_weights/Adam/iter:0"))) == 17


def test_writer_round_trip_and_h5py_reads_it(tmp_path):
    rng = np.random.default_rng(0)
    layers = [("conv2d", [("conv2d/kernel:0", rng.standard_normal((5, 5, 1, 20)).astype(np.float32)),
                          ("conv2d/bias:0", rng.standard_normal(20).astype(np.float32))]),
              ("pool", []),
              ("logits", [("logits/kernel:0", rng.standard_normal((1, 1, 50, 3)).astype(np.float32)),
                          ("logits/bias:0", np.zeros(3, np.float32))])]
    layers += [("conv2d_%d" % i, [("conv2d_%d/kernel:0" % i, rng.standard_normal((3, 3, 2, 2)).astype(np.float32))])
               for i in range(1, 41)]                            # > 32 entries: several symbol nodes
    p = str(tmp_path / "w.h5")
    h5lite.write_keras_weights(p, layers)
    back = h5lite.read_keras_weights(p)
    assert [n for n, _ in back] == [n for n, _ in layers]
    for (_, ws), (_, wb) in zip(layers, back):
        assert [n for n, _ in ws] == [n for n, _ in wb]
        assert all(np.array_equal(a, b) for (_, a), (_, b) in zip(ws, wb))
    conda = "/opt/conda/bin/python3.9"
    if not os.path.exists(conda):
        pytest.skip("no interpreter with h5py in this environment")
    code = ("import h5py,numpy as np,sys\n"
            "f=h5py.File(sys.argv[1],'r')\n"
            "names=[n.decode() if isinstance(n,bytes) else n for n in f.attrs['layer_names']]\n"
            "tot=0.0\n"
            "for n in names:\n"
            "    for w in f[n].attrs['weight_names']:\n"
            "        tot+=float(np.asarray(f[n][w],dtype=np.float64).sum())\n"
            "print(len(names), repr(tot))\n")
    r = subprocess.run([conda, "-c", code, p], capture_output=True, text=True, timeout=120)
    if r.returncode != 0 and "No module named" in r.stderr:
        pytest.skip("h5py not importable")
    assert r.returncode == 0, r.stderr
    n, tot = r.stdout.split()
    want = sum(float(a.astype(np.float64).sum()) for _, ws in layers for _, a in ws)
    assert int(n) == len(layers) and abs(float(tot) - want) < 1e-6


def test_rejects_non_hdf5_and_missing_attribute(tmp_path):
    p = tmp_path / "x.h5"
    p.write_bytes(b"not an hdf5 file" * 100)
    with pytest.raises(h5lite.H5Error):
        h5lite.read_keras_weights(str(p))
    q = str(tmp_path / "empty.h5")
    w = h5lite._Writer()
    root, bt, heap = w.group({})
    w.finish(root, bt, heap, q)
    with pytest.raises(h5lite.H5Error):
        h5lite.read_keras_weights(q)
