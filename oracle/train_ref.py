"""Floating-point reference of the train step (TEST INFRASTRUCTURE ONLY): the fcn / fcn_skip graph
in torch on the CPU (autograd gives the gradients), the reference's loss and metrics
(lib/metrics.py:8-17,60-85) and the Keras-TF2.5 Adam update with per-tensor clipnorm
(lib/network.py:90-104, lib/architecture.py:83), restated in NumPy."""
from collections import OrderedDict

import numpy as np


def fcn_loss_and_grads(arch, Wt, image_u8, mask_u8, loss_kind="categorical_crossentropy", float64=False):
    """-> (loss, acc, jaccard, dice, grads dict in Keras layouts).  loss_kind: a value of the reference's
    Loss enum (lib/metrics.py:116-121); Keras reduces the tensor a loss function returns by a plain mean.
    float64=True evaluates the same graph in double precision (full-size pages: a float32 bias gradient is a cancelling sum
    over 3.1 M pixels, and torch's own float32 reduction is then off by ~1 % -- the referee for the engine must not be)."""
    import torch
    import torch.nn.functional as F
    skip = arch == "fcn_skip"
    dt = torch.float64 if float64 else torch.float32
    T = OrderedDict((k, torch.tensor(v, dtype=dt, requires_grad=True)) for k, v in Wt.items())
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

    x = torch.from_numpy(image_u8.astype(np.float32) / np.float32(255.0))[None, None].to(dt)
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
    if loss_kind == "categorical_crossentropy":
        loss = F.cross_entropy(z.reshape(-1, C), y.reshape(-1))                  # lib/metrics.py:8-9
    else:
        oh_l = F.one_hot(y, C).to(dt)
        if loss_kind in ("dice", "jaccard", "dice_and_crossentropy"):            # :60-85,107-109
            p_l = torch.softmax(z, -1)
            I = (oh_l * p_l).abs().sum((0, 1))
            S = (oh_l + p_l).abs().sum((0, 1))
            coef = (I + 100) / (S - I + 100) if loss_kind == "jaccard" else (2.0 * I + 100) / (S + 100)
            loss = (-torch.log(coef)).mean()
            if loss_kind == "dice_and_crossentropy":                             # alpha = 1: (1*dice + 0*ce) / 2
                loss = loss / 2
        elif loss_kind == "categorical_hinge":                                   # :88-95 (on the raw logits)
            pos = (oh_l * z).sum(-1)
            neg = ((1.0 - oh_l) * z).max(-1).values
            loss = torch.clamp(neg - pos + 1, min=0.0).mean()
        elif loss_kind == "categorical_focal":                                   # :98-105 (raw logits clipped)
            pc = torch.clamp(z, 1e-7, 1.0 - 1e-7)
            loss = (-oh_l * (0.25 * (1 - pc) ** 2 * torch.log(pc))).mean() * 100
        else:
            raise ValueError(loss_kind)
    loss.backward()
    with torch.no_grad():
        acc = (z.argmax(-1) == y).float().mean().item()                          # :12-17
        p = torch.softmax(z, -1)
        oh = F.one_hot(y, C).to(dt)
        inter = (oh * p).abs().sum((0, 1))
        s = (oh + p).abs().sum((0, 1))
        jac = ((inter + 100) / (s - inter + 100)).mean().item()                  # :60-69
        dice = ((2.0 * inter + 100) / (s + 100)).mean().item()                   # :76-85
    grads = OrderedDict((k, t.grad.numpy().astype(np.float32)) for k, t in T.items())
    return float(loss.item()), acc, jac, dice, grads


def dropout_keep(n, key, rate):
    """The engine's counter-based Dropout mask (pseg_engine.hip: dropout_kernel) restated: element i is kept iff the
    top 24 bits of a 32-bit mix of (i, key) are >= rate * 2^24."""
    with np.errstate(over="ignore"):
        h = np.arange(n, dtype=np.uint32) * np.uint32(0x9E3779B1) + np.uint32(key & 0xFFFFFFFF)
        h ^= h >> np.uint32(16); h *= np.uint32(0x85EBCA6B); h ^= h >> np.uint32(13); h *= np.uint32(0xC2B2AE35); h ^= h >> np.uint32(16)
    return (h >> np.uint32(8)) >= np.uint32(int(np.float32(rate) * np.float32(16777216.0)))


def dropout_key(seed, step, op_index):
    """Key of the Dropout behind op `op_index` in training forward number `step` (0-based) after
    pseg_train_set_dropout_seed(seed)."""
    base = ((seed * 0x632BE5AB + step * 0x9E3779B9) & 0xFFFFFFFF) | 1
    return (base + 0x85EBCA77 * op_index) & 0xFFFFFFFF


def graph_loss_and_grads(arch, Wt, image_u8, mask_u8, drop=None, float64=False, bn_stats=None, route_acts=None):
    """unet (lib/model.py:151-203) and res_unet (:237-307) in torch with the reference's cross-entropy (lib/metrics.py:8-9);
    `drop` = (seed, step) enables unet's two Dropout(0.5) layers with the engine's masks (op indices 10 and 13 of the
    engine's op list; masks are laid out over the (H/8, W/8, 512) and (H/16, W/16, 1024) canvases).
    A weight table holding "batch_normalization/gamma" selects res_unet with BatchNormalization at its bn_act sites
    (lib/model.py:265-271) in TRAINING form: batch statistics (biased variance, eps 1e-3); `bn_stats` (a dict), when
    given, receives name -> (batch mean, biased batch variance, samples seen) for the moving-statistics update
    moving -= (moving - batch) * (1 - 0.99), the variance with Bessel's correction.
    `route_acts` (unet): the float32 oracle's activations (oracle.forward(..., return_acts=True)), which the engine
    reproduces bit for bit.  A 2x2 max-pool routes its gradient to ONE element of the window; where the two largest values
    of a window differ by less than float32 rounding, which one wins depends on the summation order of the convolution
    that produced them, and a referee that picks the other one moves that gradient to a neighbouring pixel (observed:
    one such window in 5 215 puts 1e-2 relative error on conv2d_5/kernel, every float32 order has its own).  With
    route_acts the referee takes each window's winner from the float32 activations (first maximum in (0,0), (0,1),
    (1,0), (1,1) order, as pool_bwd_kernel does), so that only rounding separates it from the engine.
    -> (loss, grads dict in Keras layouts, logits (H,W,C))."""
    import torch
    import torch.nn.functional as F
    dt = torch.float64 if float64 else torch.float32
    T = OrderedDict((k, torch.tensor(v, dtype=dt, requires_grad=True)) for k, v in Wt.items())
    use_bn = "batch_normalization/gamma" in Wt
    bn_names = iter(["batch_normalization"] + ["batch_normalization_%d" % i for i in range(1, 64)])
    H, W = image_u8.shape
    ph, pw = (32 - H % 32) % 32, (32 - W % 32) % 32
    names = iter(["conv2d"] + ["conv2d_%d" % i for i in range(1, 64)])

    def conv(x, n, k, relu=False, stride=1, pre_relu=False):
        if pre_relu:
            x = F.relu(x)
        Hin, Win = x.shape[2], x.shape[3]
        th = max((-(-Hin // stride) - 1) * stride + k - Hin, 0)
        tw = max((-(-Win // stride) - 1) * stride + k - Win, 0)
        x = F.pad(x, (tw // 2, tw - tw // 2, th // 2, th - th // 2))           # TF SAME
        y = F.conv2d(x, T[n + "/kernel"].permute(3, 2, 0, 1), T[n + "/bias"], stride=stride)
        return F.relu(y) if relu else y

    last_keep = [None]

    def dropped(x, op_index):
        last_keep[0] = None
        if drop is None:
            return x
        _, c, h, w = x.shape
        keep = dropout_keep(h * w * c, dropout_key(drop[0], drop[1], op_index), 0.5).reshape(h, w, c)
        last_keep[0] = keep
        m = torch.from_numpy(np.ascontiguousarray(keep.transpose(2, 0, 1)[None]).astype(np.float32))
        return x * m * 2.0

    def pool(x, layer, keep=None):
        if route_acts is None:
            return F.max_pool2d(x, 2)
        a = np.asarray(route_acts[layer], np.float32)
        if keep is not None:
            a = a * keep.astype(np.float32) * np.float32(2.0)
        h, w, c = a.shape
        win = a.reshape(h // 2, 2, w // 2, 2, c).transpose(0, 2, 4, 1, 3).reshape(h // 2, w // 2, c, 4)
        idx = torch.from_numpy(win.argmax(-1))[..., None]                     # first maximum, (0,0) (0,1) (1,0) (1,1)
        xv = x[0].permute(1, 2, 0).reshape(h // 2, 2, w // 2, 2, c).permute(0, 2, 4, 1, 3).reshape(h // 2, w // 2, c, 4)
        return torch.gather(xv, -1, idx).squeeze(-1).permute(2, 0, 1)[None]

    def bn(x, n, relu):
        if bn_stats is not None:
            xd = x.detach().double()
            bn_stats[n] = (xd.mean((0, 2, 3)).numpy(), xd.var((0, 2, 3), unbiased=False).numpy(), x.shape[0] * x.shape[2] * x.shape[3])
        y = F.batch_norm(x, None, None, T[n + "/gamma"], T[n + "/beta"], training=True, eps=1e-3)
        return F.relu(y) if relu else y

    x = torch.from_numpy(image_u8.astype(np.float32) / np.float32(255.0)).to(dt)[None, None]
    x = F.pad(x, (0, pw, 0, ph))
    up = lambda t: F.interpolate(t, scale_factor=2, mode="nearest")
    if arch == "unet":
        f = [64, 128, 256, 512, 1024]
        t, skips = x, []
        for l in range(5):
            t = conv(t, next(names), 3, True)
            second = next(names)
            t = conv(t, second, 3, True)
            last_keep[0] = None
            if l == 3: t = dropped(t, 10)
            if l == 4: t = dropped(t, 13)
            if l < 4:
                skips.append(t)
                t = pool(t, second, last_keep[0])
        for l in (3, 2, 1, 0):
            u = conv(up(t), next(names), 2, True)
            t = conv(torch.cat([skips[l], u], 1), next(names), 3, True)
            t = conv(t, next(names), 3, True)
    elif arch == "res_unet" and use_bn:
        def cblock(t, n, stride=1):                      # conv_block: bn_act -> Conv2D (creation order: BN, conv)
            return conv(bn(t, next(bn_names), True), n, 3, stride=stride)

        def residual(t, stride):
            b1_, b2_ = next(bn_names), next(bn_names)    # conv_block 1, conv_block 2, then the shortcut's
            n1, n2, n3 = next(names), next(names), next(names)
            r = conv(bn(t, b1_, True), n1, 3, stride=stride)
            r = conv(bn(r, b2_, True), n2, 3)
            return bn(conv(t, n3, 3, stride=stride), next(bn_names), False) + r
        n1 = next(names)
        b0_ = next(bn_names)
        n2, n3 = next(names), next(names)
        e1 = conv(bn(conv(x, n1, 3), b0_, True), n2, 3) + bn(conv(x, n3, 1), next(bn_names), False)
        e2 = residual(e1, 2); e3 = residual(e2, 2); e4 = residual(e3, 2); e5 = residual(e4, 2)
        b0 = cblock(e5, next(names))
        b1 = cblock(b0, next(names))
        t = b1
        for sk in (e4, e3, e2, e1):
            t = residual(torch.cat([up(t), sk], 1), 1)
    elif arch == "res_unet":
        def residual(t, stride):
            n1, n2, n3 = next(names), next(names), next(names)
            r = conv(t, n1, 3, stride=stride, pre_relu=True)
            r = conv(r, n2, 3, pre_relu=True)
            return conv(t, n3, 3, stride=stride) + r
        n1, n2, n3 = next(names), next(names), next(names)
        e1 = conv(conv(x, n1, 3), n2, 3, pre_relu=True) + conv(x, n3, 1)
        e2 = residual(e1, 2); e3 = residual(e2, 2); e4 = residual(e3, 2); e5 = residual(e4, 2)
        b0 = conv(e5, next(names), 3, pre_relu=True)
        b1 = conv(b0, next(names), 3, pre_relu=True)
        t = b1
        for sk in (e4, e3, e2, e1):
            t = residual(torch.cat([up(t), sk], 1), 1)
    else:
        raise ValueError(arch)
    t = t[:, :, :H, :W]
    z = F.conv2d(t, T["logits/kernel"].permute(3, 2, 0, 1), T["logits/bias"])[0].permute(1, 2, 0)
    y = torch.from_numpy(mask_u8.astype(np.int64))
    loss = F.cross_entropy(z.reshape(-1, z.shape[-1]), y.reshape(-1))
    loss.backward()
    grads = OrderedDict((k, (t_.grad.numpy().astype(np.float32) if t_.grad is not None else np.zeros_like(Wt[k]))) for k, t_ in T.items())
    return float(loss.item()), grads, z.detach().numpy().astype(np.float32)


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


class KerasOptimizer:
    """NumPy restatement (float64) of the TF 2.5 Keras optimizer_v2 update rules with their default
    hyper-parameters, per-tensor clip_by_norm first -- the reference's Optimizers enum
    (lib/architecture.py:71-90).  TensorFlow cannot run here: unpinned against TF itself."""

    def __init__(self, name, lr, clipnorm=1.0):
        self.name, self.lr, self.clipnorm = name, lr, clipnorm
        self.t = 0
        self.m, self.v = {}, {}
        self.m_schedule = 1.0

    def apply(self, Wt, grads):
        self.t += 1
        t, lr, b1, b2, eps = self.t, self.lr, 0.9, 0.999, 1e-7
        for k, w in Wt.items():
            g = grads[k].astype(np.float64)
            if self.clipnorm and self.clipnorm > 0:
                n = np.sqrt((g * g).sum())
                g = g * self.clipnorm / max(n, self.clipnorm)
            m = self.m.get(k, np.zeros_like(g))
            v = self.v.get(k, np.full_like(g, 0.1) if self.name == "adagrad" else np.zeros_like(g))
            if self.name == "sgd":
                upd = lr * g
            elif self.name == "rmsprop":
                v = v + (g * g - v) * (1 - 0.9)
                upd = lr * g / np.sqrt(v + eps)
            elif self.name == "adagrad":
                v = v + g * g
                upd = lr * g / (np.sqrt(v) + eps)
            elif self.name == "adadelta":
                v = 0.95 * v + 0.05 * g * g
                u = np.sqrt(m + eps) / np.sqrt(v + eps) * g
                m = 0.95 * m + 0.05 * u * u
                upd = lr * u
            elif self.name == "adamax":
                m = b1 * m + (1 - b1) * g
                v = np.maximum(b2 * v, np.abs(g))
                upd = lr / (1 - b1 ** t) * m / (v + eps)
            elif self.name == "nadam":
                if k == next(iter(Wt)):
                    self._ut = b1 * (1 - 0.5 * 0.96 ** (0.004 * t))
                    self._ut1 = b1 * (1 - 0.5 * 0.96 ** (0.004 * (t + 1)))
                    self.m_schedule = self.m_schedule * self._ut
                ms_new, ms_next = self.m_schedule, self.m_schedule * self._ut1
                gp = g / (1 - ms_new)
                m = b1 * m + (1 - b1) * g
                v = b2 * v + (1 - b2) * g * g
                mbar = (1 - self._ut) * gp + self._ut1 * (m / (1 - ms_next))
                upd = lr * mbar / (np.sqrt(v / (1 - b2 ** t)) + eps)
            elif self.name == "adam":
                m = b1 * m + (1 - b1) * g
                v = b2 * v + (1 - b2) * g * g
                upd = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t) * m / (np.sqrt(v) + eps)
            else:
                raise ValueError(self.name)
            self.m[k], self.v[k] = m, v
            Wt[k] = (w.astype(np.float64) - upd).astype(np.float32)
