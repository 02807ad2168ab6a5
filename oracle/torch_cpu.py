"""float32 restatement of the fcn / fcn_skip predict path on torch-CPU (oneDNN convolutions) -- TEST
INFRASTRUCTURE ONLY: the CPU baseline bench.py times beside the GPU path (the
reference's TensorFlow-CPU path cannot run offline, so the stand-in is the same graph on torch's CPU
backend at n = 1 and n = all cores) and a second, independent implementation the C oracle is checked
against (tests/test_oracle.py).  Graph: lib/model.py:45-92 (fcn_skip), :206-234 (fcn); predict:
lib/network.py:248-260."""
import numpy as np


def _prep(Wt):
    import torch
    out = {}
    for k, v in Wt.items():
        t = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
        # Keras Conv2D (kh,kw,Cin,Cout) -> torch (Cout,Cin,kh,kw); Conv2DTranspose (kh,kw,Cout,Cin) -> torch (Cin,Cout,kh,kw)
        out[k] = t.permute(3, 2, 0, 1).contiguous() if t.ndim == 4 else t
    return out


def fcn_forward(arch, Wt, image_u8, threads=None, prepared=None):
    """uint8 (H,W) page -> (logits float32 (H,W,C), labels int64 (H,W)); `threads` sets torch's intra-op
    thread count for this call (None: leave it)."""
    import torch
    import torch.nn.functional as F
    if arch not in ("fcn", "fcn_skip"):
        raise ValueError("torch-CPU restatement covers fcn and fcn_skip (got %r)" % (arch,))
    skip = arch == "fcn_skip"
    if threads:
        torch.set_num_threads(int(threads))
    T = prepared if prepared is not None else _prep(Wt)
    H, W = image_u8.shape
    ph, pw = (32 - H % 32) % 32, (32 - W % 32) % 32           # lib/model.py:10-26: zero pad bottom / right

    def conv(x, n, relu):
        y = F.conv2d(x, T[n + "/kernel"], T[n + "/bias"], padding=2)
        return F.relu(y) if relu else y

    def tconv5(x, n):
        return F.relu(F.conv_transpose2d(x, T[n + "/kernel"], T[n + "/bias"], padding=2))

    def dec2(x, n, relu):
        y = F.conv_transpose2d(x, T[n + "/kernel"], T[n + "/bias"], stride=2)
        return F.relu(y) if relu else y

    with torch.inference_mode():
        x = torch.from_numpy(image_u8.astype(np.float32) / np.float32(255.0))[None, None]   # lib/architecture.py:67-68
        x = F.pad(x, (0, pw, 0, ph))
        c1 = conv(x, "conv2d", True); c2 = conv(c1, "conv2d_1", False)
        c3 = conv(F.max_pool2d(c2, 2), "conv2d_2", True); c4 = conv(c3, "conv2d_3", False)
        c5 = conv(F.max_pool2d(c4, 2), "conv2d_4", True); c6 = conv(c5, "conv2d_5", False)
        c7 = conv(F.max_pool2d(c6, 2), "conv2d_6", True)
        d1 = tconv5(c7, "conv2d_transpose")
        d2 = dec2(d1, "conv2d_transpose_1", True)
        if skip: d2 = torch.cat([d2, c6], 1)
        d3 = tconv5(d2, "conv2d_transpose_2")
        if skip: d3 = torch.cat([d3, c5], 1)
        d4 = dec2(d3, "conv2d_transpose_3", True)
        if skip: d4 = torch.cat([d4, c3], 1)
        d5 = dec2(d4, "conv2d_transpose_4", False)
        if skip: d5 = torch.cat([d5, c2], 1)
        d5 = d5[:, :, :H, :W]                                  # crop, lib/model.py:29-42
        z = F.conv2d(d5, T["logits/kernel"], T["logits/bias"])[0].permute(1, 2, 0).contiguous()
        lab = z.argmax(-1)                                     # lib/network.py:259
    return z.numpy(), lab.numpy()


def time_predict(arch, Wt, image_u8, threads, warmup=3, reps=10):
    """Median seconds of `reps` predict calls after `warmup` untimed ones."""
    import time
    T = _prep(Wt)
    for _ in range(warmup):
        fcn_forward(arch, Wt, image_u8, threads=threads, prepared=T)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fcn_forward(arch, Wt, image_u8, threads=threads, prepared=T)
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2]
