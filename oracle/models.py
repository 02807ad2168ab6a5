"""Oracle model graphs (TEST INFRASTRUCTURE ONLY): eager NumPy/C restatement of the reference's
Keras constructors.  Each function cites the reference lines it follows.

Weight dict keys follow Keras' default layer naming in creation order inside a fresh session
("conv2d", "conv2d_1", ..., "conv2d_transpose", ..., "logits") with "/kernel" and "/bias"
suffixes; kernel layouts are Keras': Conv2D (kh,kw,Cin,Cout), Conv2DTranspose (kh,kw,Cout,Cin).

mode="f32":  every tensor float32, sequential-fmaf accumulation (oracle/pseg_oracle.c).
mode="bf16": build-defined throughput mode -- kernels rounded to bf16 once, every layer output
             (after bias / ReLU / residual add) rounded to bf16, accumulation still float32,
             final logits float32.  This is what the MFMA path is compared with.
"""
from collections import OrderedDict

import numpy as np

from . import core

ARCHS = ("fcn_skip", "fcn", "unet", "res_unet")


class _Namer:
    """Keras auto-naming: per-class counter, first instance has no suffix."""

    def __init__(self):
        self.n = {}

    def __call__(self, base):
        i = self.n.get(base, 0)
        self.n[base] = i + 1
        return base if i == 0 else "%s_%d" % (base, i)


BN_EPS = np.float32(1e-3)     # tf.keras.layers.BatchNormalization defaults (lib/model.py:268 passes none)
BN_MOMENTUM = np.float32(0.99)


def weight_specs(arch, n_classes, in_ch=1, batch_norm=False):
    """[(layer_name, kind, kernel_shape_keras, cout)] in Keras creation order.
    kind: 'conv' | 'tconv' (Conv2DTranspose) | 'bn' (BatchNormalization over `cout` channels; only with batch_norm=True,
    at res_unet's bn_act sites, lib/model.py:265-271)."""
    nm = _Namer()
    S = []

    def bn(c):
        if batch_norm:
            S.append((nm("batch_normalization"), "bn", (c,), c))

    def conv(cin, cout, k, name=None):
        S.append((name or nm("conv2d"), "conv", (k, k, cin, cout), cout))

    def tconv(cin, cout, k):
        S.append((nm("conv2d_transpose"), "tconv", (k, k, cout, cin), cout))

    if arch in ("fcn_skip", "fcn"):
        # lib/model.py:50-66 / :211-220 (encoder identical)
        for cin, cout in ((in_ch, 20), (20, 30), (30, 40), (40, 40), (40, 60), (60, 60), (60, 80)):
            conv(cin, cout, 5)
        skip = arch == "fcn_skip"
        tconv(80, 80, 5)                       # deconv1  :69 / :223
        tconv(80, 60, 2)                       # deconv2  :71 / :224
        tconv(120 if skip else 60, 40, 5)      # deconv3  :75 / :226
        tconv(100 if skip else 40, 30, 2)      # deconv4  :79 / :227
        tconv(70 if skip else 30, 20, 2)       # deconv5  :83 / :229
        conv(50 if skip else 20, n_classes, 1, "logits")  # :88 / :231
    elif arch == "unet":
        # lib/model.py:156-199
        c = in_ch
        for f in (64, 128, 256, 512, 1024):
            conv(c, f, 3)
            conv(f, f, 3)
            c = f
        for f in (512, 256, 128, 64):
            conv(2 * f, f, 2)      # up-conv k2 on the upsampled tensor
            conv(2 * f, f, 3)      # on concat [skip, up]
            conv(f, f, 3)
        conv(64, n_classes, 1, "logits")
    elif arch == "res_unet":
        # lib/model.py:237-307, f = [32,64,128,256,512]
        f = [32, 64, 128, 256, 512]
        conv(in_ch, f[0], 3)       # stem conv            :252
        bn(f[0])
        conv(f[0], f[0], 3)        # stem conv_block      :253
        conv(in_ch, f[0], 1)       # stem shortcut k1     :254
        bn(f[0])                   # bn_act(shortcut, act=False) :255

        def res(cin, cout):
            bn(cin)
            conv(cin, cout, 3)     # conv_block 1 (stride s)   :244
            bn(cout)
            conv(cout, cout, 3)    # conv_block 2              :245
            conv(cin, cout, 3)     # shortcut (stride s)       :246
            bn(cout)               # bn_act(shortcut, act=False) :247

        res(f[0], f[1]); res(f[1], f[2]); res(f[2], f[3]); res(f[3], f[4])   # :281-284
        bn(f[4]); conv(f[4], f[4], 3); bn(f[4]); conv(f[4], f[4], 3)         # bridge :287-288
        res(f[4] + f[3], f[4])     # d1 :291-292
        res(f[4] + f[2], f[3])     # d2 :294-295
        res(f[3] + f[1], f[2])     # d3 :297-298
        res(f[2] + f[0], f[1])     # d4 :300-301
        conv(f[1], n_classes, 1, "logits")
    else:
        raise ValueError(arch)
    return S


def init_weights(arch, n_classes, seed=42, in_ch=1, gain=1.0, bias_scale=0.0, batch_norm=False):
    """Synthetic weights: Keras glorot_uniform limits, numpy default_rng(seed) in
    layer order; biases zero unless bias_scale>0 (then U(-bias_scale, bias_scale), so that the
    pad-to-32 region and bias paths are exercised)."""
    rng = np.random.default_rng(seed)
    Wt = OrderedDict()
    for name, kind, shp, cout in weight_specs(arch, n_classes, in_ch, batch_norm):
        if kind == "bn":   # Keras: ones / zeros / zeros / ones; perturbed when bias_scale > 0 so that every term is exercised
            j = bias_scale > 0
            Wt[name + "/gamma"] = (1.0 + j * rng.uniform(-0.3, 0.3, size=shp)).astype(np.float32)
            Wt[name + "/beta"] = (j * rng.uniform(-bias_scale, bias_scale, size=shp)).astype(np.float32)
            Wt[name + "/moving_mean"] = (j * rng.uniform(-bias_scale, bias_scale, size=shp)).astype(np.float32)
            Wt[name + "/moving_variance"] = (1.0 + j * rng.uniform(-0.3, 0.3, size=shp)).astype(np.float32)
            continue
        rf = shp[0] * shp[1]
        limit = np.sqrt(6.0 / (rf * shp[2] + rf * shp[3])) * gain
        Wt[name + "/kernel"] = rng.uniform(-limit, limit, size=shp).astype(np.float32)
        if bias_scale > 0:
            Wt[name + "/bias"] = rng.uniform(-bias_scale, bias_scale, size=(cout,)).astype(np.float32)
        else:
            Wt[name + "/bias"] = np.zeros((cout,), np.float32)
    return Wt


def to_bf16_weights(Wt):
    """Kernels rounded to bf16 (biases stay f32)."""
    return OrderedDict((k, core.round_bf16(v) if k.endswith("/kernel") else v) for k, v in Wt.items())


def _pad32(x):
    """lib/model.py:10-26: zero-pad bottom/right to a multiple of 32."""
    H, W, _ = x.shape
    ph, pw = (32 - H % 32) % 32, (32 - W % 32) % 32
    return np.pad(x, ((0, ph), (0, pw), (0, 0))), (ph, pw)


def _crop(x, pads):
    """lib/model.py:29-42."""
    ph, pw = pads
    H, W, _ = x.shape
    return np.ascontiguousarray(x[:H - ph, :W - pw])


def _up2(x):
    """UpSampling2D(2), nearest (lib/model.py:175,239)."""
    return np.repeat(np.repeat(x, 2, axis=0), 2, axis=1)


class _Ctx:
    def __init__(self, Wt, mode):
        self.W = to_bf16_weights(Wt) if mode == "bf16" else Wt
        self.q = core.round_bf16 if mode == "bf16" else (lambda a: a)
        self.bf16 = mode == "bf16"
        self.nm = _Namer()
        self.acts = OrderedDict()

    def conv(self, x, relu=False, stride=1, name=None, add=None, in_relu=False, keep=None):
        name = name or self.nm("conv2d")
        if in_relu:
            x = np.maximum(x, 0)
        y = core.conv2d(x, self.W[name + "/kernel"], self.W[name + "/bias"], stride=stride,
                        relu=relu and add is None)
        if add is not None:
            y = y + add
            if relu:
                y = np.maximum(y, 0)
        if name != "logits":
            y = self.q(y)
        self.acts[keep or name] = y
        return y

    def bn(self, x, name, relu=False):
        """BatchNormalization at inference (moving statistics), then bn_act's optional ReLU (lib/model.py:265-271).
        f32: y = (x - mean) * (gamma / sqrt(var + eps)) + beta; bf16 engine: per-channel scale / shift in float32 on
        the bf16 tensor."""
        g, b = self.W[name + "/gamma"], self.W[name + "/beta"]
        m, v = self.W[name + "/moving_mean"], self.W[name + "/moving_variance"]
        scale = (g / np.sqrt(v + BN_EPS)).astype(np.float32)
        if self.bf16:
            y = x * scale + (b - m * scale).astype(np.float32)
        else:
            y = (x - m) * scale + b
        if relu:
            y = np.maximum(y, 0)
        y = self.q(y.astype(np.float32))
        self.acts[name] = y
        return y

    def tconv5(self, x, relu=False):
        """Conv2DTranspose k5 s1 SAME == correlation with the flipped, channel-swapped kernel."""
        name = self.nm("conv2d_transpose")
        K = self.W[name + "/kernel"]                       # (5,5,Cout,Cin)
        wc = np.ascontiguousarray(np.transpose(K[::-1, ::-1], (0, 1, 3, 2)))
        y = self.q(core.conv2d(x, wc, self.W[name + "/bias"], relu=relu))
        self.acts[name] = y
        return y

    def deconv2(self, x, relu=False):
        name = self.nm("conv2d_transpose")
        K = self.W[name + "/kernel"]                       # (2,2,Cout,Cin)
        w = np.ascontiguousarray(np.transpose(K, (0, 1, 3, 2)))
        y = self.q(core.deconv2x2(x, w, self.W[name + "/bias"], relu=relu))
        self.acts[name] = y
        return y


def _cat(a, b):
    return np.ascontiguousarray(np.concatenate([a, b], axis=-1))


def _fcn(c, x, skip):
    """lib/model.py:45-92 (skip=True) / :206-234 (skip=False)."""
    x, pads = _pad32(x)
    c1 = c.conv(x, relu=True)
    c2 = c.conv(c1)
    c3 = c.conv(core.maxpool2(c2), relu=True)
    c4 = c.conv(c3)
    c5 = c.conv(core.maxpool2(c4), relu=True)
    c6 = c.conv(c5)
    c7 = c.conv(core.maxpool2(c6), relu=True)
    d1 = c.tconv5(c7, relu=True)
    d2 = c.deconv2(d1, relu=True)
    if skip:
        d2 = _cat(d2, c6)          # [deconv, skip] order, :73
    d3 = c.tconv5(d2, relu=True)
    if skip:
        d3 = _cat(d3, c5)          # :77
    d4 = c.deconv2(d3, relu=True)
    if skip:
        d4 = _cat(d4, c3)          # :81
    d5 = c.deconv2(d4)
    if skip:
        d5 = _cat(d5, c2)          # :85
    d5 = _crop(d5, pads)           # :86
    return c.conv(d5, name="logits")


def _unet(c, x):
    """lib/model.py:151-203.  Dropout is the identity at inference."""
    x, pads = _pad32(x)
    skips = []
    t = x
    for lvl in range(5):
        t = c.conv(t, relu=True)
        t = c.conv(t, relu=True)
        if lvl < 4:
            skips.append(t)
            t = core.maxpool2(t)
    for lvl in range(4):
        up = c.conv(_up2(t), relu=True)            # k2 SAME: pad 0 before / 1 after
        t = _cat(skips[3 - lvl], up)               # [skip, up] order, :176
        t = c.conv(t, relu=True)
        t = c.conv(t, relu=True)
    t = _crop(t, pads)
    return c.conv(t, name="logits")


def _res_unet(c, x, bn=False):
    """lib/model.py:237-307.  bn=False is the reference as shipped (BatchNorm flag hard-wired off, :265); bn=True places
    BatchNormalization where bn_act would, names in Keras creation order."""
    x, pads = _pad32(x)
    bnn = (lambda: c.nm("batch_normalization")) if bn else (lambda: None)

    def pre(t, name):
        """bn_act(t): [BatchNormalization,] ReLU -- returns (tensor, whether the conv still has to apply the ReLU)"""
        return (c.bn(t, name, relu=True), False) if bn else (t, True)

    def residual(t, stride=1):
        a, ir = pre(t, bnn())
        r = c.conv(a, stride=stride, in_relu=ir)            # conv_block 1
        # Keras creation order: [bn] res conv1, [bn] res conv2, shortcut [bn] -- allocate names in that order
        bn2 = bnn()
        n2 = c.nm("conv2d")
        sc_name = c.nm("conv2d")
        sc = c.conv(t, stride=stride, name=sc_name)         # shortcut, no activation
        if bn:
            sc = c.bn(sc, bnn())
        a2, ir2 = pre(r, bn2)
        return c.conv(a2, in_relu=ir2, name=n2, add=sc)     # Add()([shortcut, res])

    # stem :251-257
    s = c.conv(x)
    bn0 = bnn()
    n2 = c.nm("conv2d")
    sc = c.conv(x, name=c.nm("conv2d"))                     # k1 shortcut
    if bn:
        sc = c.bn(sc, bnn())
    a, ir = pre(s, bn0)
    e1 = c.conv(a, in_relu=ir, name=n2, add=sc)
    e2 = residual(e1, 2)
    e3 = residual(e2, 2)
    e4 = residual(e3, 2)
    e5 = residual(e4, 2)
    a, ir = pre(e5, bnn())
    b0 = c.conv(a, in_relu=ir)
    a, ir = pre(b0, bnn())
    b1 = c.conv(a, in_relu=ir)
    d = b1
    for sk in (e4, e3, e2, e1):
        d = residual(_cat(_up2(d), sk))                     # [up, skip] order, :240
    d = _crop(d, pads)
    return c.conv(d, name="logits")


def forward(arch, Wt, image_u8, mode="f32", return_acts=False):
    """uint8 (H,W) network input (already inverted / line-height normalised) -> logits (H,W,C) f32.
    lib/network.py:250-257: preprocess = x/255.0, batch of one."""
    img = np.asarray(image_u8)
    assert img.dtype == np.uint8 and img.ndim in (2, 3)      # (H,W) gray or (H,W,3) (input_image_dimension = 3, lib/network.py:28,56)
    c = _Ctx(Wt, mode)
    x = c.q(core.preprocess(img))
    if img.ndim == 2:
        x = x[..., None]
    if arch == "fcn_skip":
        z = _fcn(c, x, True)
    elif arch == "fcn":
        z = _fcn(c, x, False)
    elif arch == "unet":
        z = _unet(c, x)
    elif arch == "res_unet":
        z = _res_unet(c, x, bn="batch_normalization/gamma" in Wt)
    else:
        raise ValueError(arch)
    return (z, c.acts) if return_acts else z


def predict_single_data(arch, Wt, image_u8, mode="f32"):
    """lib/network.py:248-260 -> (logit f32, prob f32, pred int64).  softmax is the reference's
    own call (scipy.special.softmax on the float32 logits); argmax first-max-wins."""
    from scipy.special import softmax
    logit = forward(arch, Wt, image_u8, mode)
    prob = softmax(logit, -1)
    pred = np.argmax(logit, -1)
    return logit, prob, pred
