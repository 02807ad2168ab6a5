"""Floating-point reference of the train step (TEST INFRASTRUCTURE ONLY): the fcn / fcn_skip graph
in torch on the CPU (autograd gives the gradients), the reference's loss and metrics
(lib/metrics.py:8-17,60-85) and the Keras-TF2.5 Adam update with per-tensor clipnorm
(lib/network.py:90-104, lib/architecture.py:83), restated in NumPy."""
from collections import OrderedDict

import numpy as np


def fcn_loss_and_grads(arch, Wt, image_u8, mask_u8):
    """-> (loss, acc, jaccard, dice, grads dict in Keras layouts)."""
    import torch
    import torch.nn.functional as F
    skip = arch == "fcn_skip"
    T = OrderedDict((k, torch.tensor(v, dtype=torch.float32, requires_grad=True)) for k, v in Wt.items())
    H, W = image_u8.shape
    ph, pw = (32 - H % 32) % 32, (32 - W % 32) % 32

    def conv(x, n, relu):
        y = F.conv2d(x, T[n + "/kernel"].permute(3, 2, 0, 1), T[n + "/bias"], padding=2)
        return F.relu(y) if relu else y

    def tconv5(x, n):
        return F.relu(F.conv_transpose2d(x, T[n + "/kernel"].permute(3, 2, 0, 1), T[n + "/bias"], padding=2))

    def dec2(x, n, relu):
        y = F.conv_transpose2d(x, T[n + "/kernel"].permute(3, 2, 0, 1), T[n + "/bias"], stride=2)
        return F.relu(y) if relu else y

    x = torch.from_numpy(image_u8.astype(np.float32) / np.float32(255.0))[None, None]
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
    d5 = d5[:, :, :H, :W]
    z = F.conv2d(d5, T["logits/kernel"].permute(3, 2, 0, 1), T["logits/bias"])[0].permute(1, 2, 0)   # (H,W,C)
    y = torch.from_numpy(mask_u8.astype(np.int64))
    C = z.shape[-1]
    loss = F.cross_entropy(z.reshape(-1, C), y.reshape(-1))                      # lib/metrics.py:8-9
    loss.backward()
    with torch.no_grad():
        acc = (z.argmax(-1) == y).float().mean().item()                          # :12-17
        p = torch.softmax(z, -1)
        oh = F.one_hot(y, C).float()
        inter = (oh * p).abs().sum((0, 1))
        s = (oh + p).abs().sum((0, 1))
        jac = ((inter + 100) / (s - inter + 100)).mean().item()                  # :60-69
        dice = ((2.0 * inter + 100) / (s + 100)).mean().item()                   # :76-85
    grads = OrderedDict((k, t.grad.numpy().copy()) for k, t in T.items())
    return float(loss.item()), acc, jac, dice, grads


class KerasAdam:
    """TF 2.5 Keras Adam with per-tensor clip_by_norm applied first."""

    def __init__(self, lr, clipnorm=1.0, beta1=0.9, beta2=0.999, eps=1e-7):
        self.lr, self.clipnorm, self.b1, self.b2, self.eps = lr, clipnorm, beta1, beta2, eps
        self.t = 0
        self.m, self.v = {}, {}

    def apply(self, Wt, grads):
        self.t += 1
        lr_t = self.lr * np.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        out = OrderedDict()
        for k, w in Wt.items():
            g = grads[k].astype(np.float64)
            if self.clipnorm and self.clipnorm > 0:
                n = np.sqrt((g * g).sum())
                g = g * self.clipnorm / max(n, self.clipnorm)
            m = self.b1 * self.m.get(k, 0.0) + (1 - self.b1) * g
            v = self.b2 * self.v.get(k, 0.0) + (1 - self.b2) * g * g
            self.m[k], self.v[k] = m, v
            out[k] = (w.astype(np.float64) - lr_t * m / (np.sqrt(v) + self.eps)).astype(np.float32)
        return out
