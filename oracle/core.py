"""ctypes binding of oracle/liboracle.so (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib_path():
    return os.path.join(_HERE, "liboracle.so")


def build(force=False):
    """Compile liboracle.so with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "pseg_oracle.c")
    out = lib_path()
    if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return out


def _lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(lib_path()):
            build()
        L = ctypes.CDLL(lib_path())
        f32p = ctypes.POINTER(ctypes.c_float)
        i = ctypes.c_int
        L.orc_conv2d.argtypes = [f32p, i, i, i, f32p, f32p, i, i, i, i, i, i, i, i, i, f32p]
        L.orc_conv2d.restype = i
        L.orc_deconv2x2.argtypes = [f32p, i, i, i, f32p, f32p, i, i, f32p]
        L.orc_deconv2x2.restype = i
        L.orc_maxpool2.argtypes = [f32p, i, i, i, f32p]
        L.orc_maxpool2.restype = i
        L.orc_round_bf16.argtypes = [f32p, ctypes.c_int64]
        L.orc_preprocess.argtypes = [ctypes.POINTER(ctypes.c_uint8), ctypes.c_int64, f32p]
        L.orc_argmax.argtypes = [f32p, ctypes.c_int64, i, ctypes.POINTER(ctypes.c_int64)]
        L.orc_set_num_threads.argtypes = [i]
        L.orc_num_threads.restype = i
        _LIB = L
    return _LIB


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def set_num_threads(n):
    _lib().orc_set_num_threads(int(n))


def num_threads():
    return int(_lib().orc_num_threads())


def same_pad(n_in, k, stride):
    """TensorFlow 'SAME': (n_out, pad_before).  pad_after = pad_total - pad_before."""
    n_out = -(-n_in // stride)
    total = max((n_out - 1) * stride + k - n_in, 0)
    return n_out, total // 2


def conv2d(x, w, b, stride=1, relu=False, pad=None):
    """x (H,W,Cin) f32; w (KH,KW,Cin,Cout) correlation form; TF SAME unless pad=(pt,pl,Hout,Wout)."""
    x = _f32(x)
    w = _f32(w)
    H, W, Cin = x.shape
    KH, KW, Ci2, Cout = w.shape
    assert Ci2 == Cin, (w.shape, x.shape)
    if pad is None:
        Ho, pt = same_pad(H, KH, stride)
        Wo, pl = same_pad(W, KW, stride)
    else:
        pt, pl, Ho, Wo = pad
    out = np.empty((Ho, Wo, Cout), np.float32)
    bb = _f32(b) if b is not None else None
    rc = _lib().orc_conv2d(_p(x), H, W, Cin, _p(w), _p(bb) if bb is not None else None, KH, KW,
                           stride, pt, pl, Ho, Wo, Cout, int(bool(relu)), _p(out))
    assert rc == 0
    return out


def deconv2x2(x, w_abio, b, relu=False):
    """x (H,W,Cin); w_abio (2,2,Cin,Cout) (already transposed from Keras (2,2,Cout,Cin))."""
    x = _f32(x)
    w = _f32(w_abio)
    H, W, Cin = x.shape
    assert w.shape[:3] == (2, 2, Cin)
    Cout = w.shape[3]
    out = np.empty((2 * H, 2 * W, Cout), np.float32)
    bb = _f32(b) if b is not None else None
    rc = _lib().orc_deconv2x2(_p(x), H, W, Cin, _p(w), _p(bb) if bb is not None else None, Cout,
                              int(bool(relu)), _p(out))
    assert rc == 0
    return out


def maxpool2(x):
    x = _f32(x)
    H, W, C = x.shape
    out = np.empty((H // 2, W // 2, C), np.float32)
    rc = _lib().orc_maxpool2(_p(x), H, W, C, _p(out))
    assert rc == 0
    return out


def round_bf16(x):
    """Round-to-nearest-even to bfloat16 precision, returned as float32 (copy)."""
    y = np.array(x, dtype=np.float32, copy=True, order="C")
    _lib().orc_round_bf16(_p(y), y.size)
    return y


def preprocess(img_u8):
    a = np.ascontiguousarray(img_u8, dtype=np.uint8)
    out = np.empty(a.shape, np.float32)
    _lib().orc_preprocess(a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), a.size, _p(out))
    return out


def argmax(logits):
    z = _f32(logits)
    C = z.shape[-1]
    out = np.empty(z.shape[:-1], np.int64)
    _lib().orc_argmax(_p(z), out.size, C, out.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
    return out
