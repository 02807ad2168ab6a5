// pseg_graph.cpp -- static layer graphs of the in-scope architectures.
//
// Each builder restates one Keras constructor of the reference as a flat op list over NHWC
// tensors on the pad-to-32 canvas (lib/model.py:10-26).  Concatenate / UpSampling2D / Add /
// pre-activation ReLU are never materialised: they are gather/epilogue flags on the consuming
// or producing conv.  Layer names reproduce Keras' default naming so that weight names match
// what model.get_weights()/an .h5 file would carry.
#include <map>

#include "pseg_common.h"

namespace pseg {

namespace {

struct Builder {
    Engine& e;
    std::map<std::string, int> counters;
    explicit Builder(Engine& eng) : e(eng) {}

    std::string nm(const std::string& base) {
        int i = counters[base]++;
        return i == 0 ? base : base + "_" + std::to_string(i);
    }
    int tensor(const std::string& name, int s, int C) {
        Tensor t;
        t.name = name;
        t.s = s;
        t.C = C;
        t.Cs = (e.mode == PSEG_MODE_BF16) ? round_up(C, 8) : C;
        e.tensors.push_back(t);
        return (int)e.tensors.size() - 1;
    }
    // Reserve the weight-table slots of a layer whose op is created later than its Keras
    // name (residual blocks): the table stays in Keras creation order.
    std::string reserve(const std::string& base) {
        std::string name = nm(base);
        for (const char* suf : {"/kernel", "/bias"}) {
            Param p;
            p.name = name + suf;
            e.params.push_back(p);
        }
        return name;
    }
    int param(const std::string& name, std::initializer_list<int64_t> shp) {
        int idx = -1;
        for (size_t j = 0; j < e.params.size(); ++j)
            if (e.params[j].name == name) idx = (int)j;
        if (idx < 0) {
            Param p;
            p.name = name;
            e.params.push_back(p);
            idx = (int)e.params.size() - 1;
        }
        Param& p = e.params[idx];
        p.ndim = (int)shp.size();
        int i = 0;
        int64_t n = 1;
        for (auto v : shp) { p.shape[i++] = v; n *= v; }
        p.host.assign((size_t)n, 0.0f);
        return idx;
    }
    // BatchNormalization layer: weight-table slots in Keras order (gamma, beta, moving_mean, moving_variance)
    std::string reserve_bn() {
        std::string name = nm("batch_normalization");
        for (const char* suf : {"/gamma", "/beta", "/moving_mean", "/moving_variance"}) {
            Param p;
            p.name = name + suf;
            e.params.push_back(p);
        }
        return name;
    }
    // ... applied to one source tensor (channels [c0, c0 + C) of a layer over `ctot` channels), optionally followed
    // by the Activation('relu') of bn_act (lib/model.py:265-271)
    int bn(int src, const std::string& layer, int c0, int ctot, bool relu, int up = 0) {
        Op op;
        op.type = OP_BN;
        op.up0 = up;   // the layer sees this tensor through UpSampling2D(2)^up: same batch statistics, 4^up times the sample count
        op.layer = c0 == 0 ? layer : layer + "+" + std::to_string(c0);
        op.src0 = src;
        op.relu = relu;
        op.bn_c0 = c0;
        op.Cin = op.Cout = e.tensors[src].C;
        op.kparam = param(layer + "/gamma", {ctot});
        op.bparam = param(layer + "/beta", {ctot});
        op.mmparam = param(layer + "/moving_mean", {ctot});
        op.mvparam = param(layer + "/moving_variance", {ctot});
        op.dst = tensor(op.layer, e.tensors[src].s, op.Cin);
        e.ops.push_back(op);
        return op.dst;
    }
    int cin_of(int src0, int src1) const {
        return e.tensors[src0].C + (src1 >= 0 ? e.tensors[src1].C : 0);
    }

    // Conv2D (lib/model.py:50 etc.).  SAME padding; stride 1 or 2.
    int conv(int src0, int src1, int cout, int k, bool relu, int stride = 1,
             const std::string& name = "", bool in_relu = false, int add = -1, int up0 = 0,
             int up1 = 0) {
        Op op;
        op.type = OP_CONV;
        op.layer = name.empty() ? nm("conv2d") : name;
        op.k = k;
        op.stride = stride;
        op.src0 = src0;
        op.src1 = src1;
        op.up0 = up0;
        op.up1 = up1;
        op.in_relu = in_relu;
        op.relu = relu;
        op.add = add;
        op.Cin = cin_of(src0, src1);
        op.Cout = cout;
        int s_in = e.tensors[src0].s - up0;
        int s_out = s_in + (stride == 2 ? 1 : 0);
        op.dst = tensor(op.layer, s_out, cout);
        op.kparam = param(op.layer + "/kernel", {k, k, op.Cin, cout});
        op.bparam = param(op.layer + "/bias", {cout});
        op.flops_per_canvas_px = 2.0 * k * k * op.Cin * cout / (double)(1 << (2 * s_out));
        e.ops.push_back(op);
        return op.dst;
    }
    // Conv2DTranspose k5 s1 SAME (lib/model.py:69,75): correlation with the flipped kernel.
    int tconv5(int src0, int src1, int cout, bool relu) {
        Op op;
        op.type = OP_CONV;
        op.layer = nm("conv2d_transpose");
        op.k = 5;
        op.transposed = true;
        op.src0 = src0;
        op.src1 = src1;
        op.relu = relu;
        op.Cin = cin_of(src0, src1);
        op.Cout = cout;
        int s = e.tensors[src0].s;
        op.dst = tensor(op.layer, s, cout);
        op.kparam = param(op.layer + "/kernel", {5, 5, cout, op.Cin});
        op.bparam = param(op.layer + "/bias", {cout});
        op.flops_per_canvas_px = 2.0 * 25 * op.Cin * cout / (double)(1 << (2 * s));
        e.ops.push_back(op);
        return op.dst;
    }
    // Conv2DTranspose k2 s2 SAME (lib/model.py:71,79,83).
    int deconv2(int src0, int src1, int cout, bool relu) {
        Op op;
        op.type = OP_DECONV2;
        op.layer = nm("conv2d_transpose");
        op.k = 2;
        op.stride = 2;
        op.transposed = true;
        op.src0 = src0;
        op.src1 = src1;
        op.relu = relu;
        op.Cin = cin_of(src0, src1);
        op.Cout = cout;
        int s = e.tensors[src0].s - 1;
        op.dst = tensor(op.layer, s, cout);
        op.kparam = param(op.layer + "/kernel", {2, 2, cout, op.Cin});
        op.bparam = param(op.layer + "/bias", {cout});
        op.flops_per_canvas_px = 2.0 * op.Cin * cout / (double)(1 << (2 * s));
        e.ops.push_back(op);
        return op.dst;
    }
    // MaxPooling2D 2x2 s2 (lib/model.py:54): canvas dims are multiples of 32, so SAME == VALID.
    int pool(int src) {
        Op op;
        op.type = OP_POOL;
        op.layer = nm("max_pooling2d");
        op.src0 = src;
        op.Cin = op.Cout = e.tensors[src].C;
        op.dst = tensor(op.layer, e.tensors[src].s + 1, op.Cin);
        e.ops.push_back(op);
        return op.dst;
    }
    // crop (lib/model.py:29-42) + logits 1x1 (lib/model.py:88) + softmax/argmax
    // (lib/network.py:258-259).
    void logits(int src0, int src1) {
        Op op;
        op.type = OP_LOGITS;
        op.layer = "logits";
        op.k = 1;
        op.src0 = src0;
        op.src1 = src1;
        op.Cin = cin_of(src0, src1);
        op.Cout = e.n_classes;
        op.kparam = param("logits/kernel", {1, 1, op.Cin, e.n_classes});
        op.bparam = param("logits/bias", {e.n_classes});
        op.flops_per_canvas_px = 2.0 * op.Cin * e.n_classes;
        e.ops.push_back(op);
    }
};

// lib/model.py:45-92 (skip) and :206-234 (no skip).
void build_fcn(Engine& e, bool skip) {
    Builder b(e);
    int x = b.tensor("input", 0, e.in_ch);
    e.input_tensor = x;
    int c1 = b.conv(x, -1, 20, 5, true);
    int c2 = b.conv(c1, -1, 30, 5, false);
    int p2 = b.pool(c2);
    int c3 = b.conv(p2, -1, 40, 5, true);
    int c4 = b.conv(c3, -1, 40, 5, false);
    int p4 = b.pool(c4);
    int c5 = b.conv(p4, -1, 60, 5, true);
    int c6 = b.conv(c5, -1, 60, 5, false);
    int p6 = b.pool(c6);
    int c7 = b.conv(p6, -1, 80, 5, true);
    int d1 = b.tconv5(c7, -1, 80, true);
    int d2 = b.deconv2(d1, -1, 60, true);
    int d3 = b.tconv5(d2, skip ? c6 : -1, 40, true);   // concat [deconv2, conv6]  :73
    int d4 = b.deconv2(d3, skip ? c5 : -1, 30, true);  // concat [deconv3, conv5]  :77
    int d5 = b.deconv2(d4, skip ? c3 : -1, 20, false); // concat [deconv4, conv3]  :81
    b.logits(d5, skip ? c2 : -1);                      // concat [deconv5, conv2]  :85
}

// lib/model.py:151-203.  Dropout(0.5) is the identity at inference and live in the train step.
void build_unet(Engine& e) {
    Builder b(e);
    int t = b.tensor("input", 0, e.in_ch);
    e.input_tensor = t;
    int skips[4];
    const int f[5] = {64, 128, 256, 512, 1024};
    for (int l = 0; l < 5; ++l) {
        t = b.conv(t, -1, f[l], 3, true);
        t = b.conv(t, -1, f[l], 3, true);
        if (l >= 3) e.ops.back().dropout = 0.5f;   // drop4 / drop5 (:167,172): every consumer sees the dropped tensor
        if (l < 4) {
            skips[l] = t;
            t = b.pool(t);
        }
    }
    for (int l = 3; l >= 0; --l) {
        int up = b.conv(t, -1, f[l], 2, true, 1, "", false, -1, /*up0=*/1);  // UpSampling2D + k2
        t = b.conv(skips[l], up, f[l], 3, true);                             // [skip, up] :176
        t = b.conv(t, -1, f[l], 3, true);
    }
    b.logits(t, -1);
}

// lib/model.py:237-307.  The reference hard-wires bn_act's BatchNormalization flag to False (:265); with
// PSEG_FLAG_BATCHNORM the builder places the layer where bn_act would (in front of every pre-activation ReLU and behind
// every shortcut convolution), in Keras creation order.
void build_res_unet(Engine& e) {
    Builder b(e);
    const bool bnf = (e.flags & PSEG_FLAG_BATCHNORM) != 0;
    const int f[5] = {32, 64, 128, 256, 512};
    int x = b.tensor("input", 0, e.in_ch);
    e.input_tensor = x;
    // bn_act(x) of conv_block over the (virtual) concat [s0, s1]: one op per source, the ReLU inside the op
    struct Pre { int s0, s1, in_relu; };
    auto pre_act = [&](int s0, int s1, const std::string& layer, int up0 = 0) -> Pre {
        if (!bnf) return Pre{s0, s1, 1};
        const int ctot = b.cin_of(s0, s1);
        const int n0 = b.bn(s0, layer, 0, ctot, true, up0);
        const int n1 = s1 >= 0 ? b.bn(s1, layer, e.tensors[s0].C, ctot, true) : -1;
        return Pre{n0, n1, 0};
    };
    auto shortcut_bn = [&](int sc, const std::string& layer) { return bnf ? b.bn(sc, layer, 0, e.tensors[sc].C, false) : sc; };
    // residual_block: names allocated in Keras creation order (conv_block1, conv_block2, shortcut)
    auto residual = [&](int s0, int s1, int up0, int filters, int stride) {
        const Pre p1 = pre_act(s0, s1, bnf ? b.reserve_bn() : "", up0);
        int r = b.conv(p1.s0, p1.s1, filters, 3, false, stride, "", /*in_relu=*/p1.in_relu, -1, up0, 0);
        const std::string bn2 = bnf ? b.reserve_bn() : "";
        std::string n2 = b.reserve("conv2d");
        std::string nsc = b.reserve("conv2d");
        const std::string bnsc = bnf ? b.reserve_bn() : "";
        int sc = shortcut_bn(b.conv(s0, s1, filters, 3, false, stride, nsc, false, -1, up0, 0), bnsc);
        const Pre p2 = pre_act(r, -1, bn2);
        return b.conv(p2.s0, -1, filters, 3, false, 1, n2, /*in_relu=*/p2.in_relu, /*add=*/sc);
    };
    // stem :251-257
    int s = b.conv(x, -1, f[0], 3, false);
    const std::string bn0 = bnf ? b.reserve_bn() : "";
    std::string n2 = b.reserve("conv2d");
    int sc = b.conv(x, -1, f[0], 1, false, 1, b.reserve("conv2d"));
    sc = shortcut_bn(sc, bnf ? b.reserve_bn() : "");
    const Pre ps = pre_act(s, -1, bn0);
    int e1 = b.conv(ps.s0, -1, f[0], 3, false, 1, n2, ps.in_relu, sc);
    int e2 = residual(e1, -1, 0, f[1], 2);
    int e3 = residual(e2, -1, 0, f[2], 2);
    int e4 = residual(e3, -1, 0, f[3], 2);
    int e5 = residual(e4, -1, 0, f[4], 2);
    const Pre pb0 = pre_act(e5, -1, bnf ? b.reserve_bn() : "");
    int b0 = b.conv(pb0.s0, -1, f[4], 3, false, 1, "", pb0.in_relu);
    const Pre pb1 = pre_act(b0, -1, bnf ? b.reserve_bn() : "");
    int b1 = b.conv(pb1.s0, -1, f[4], 3, false, 1, "", pb1.in_relu);
    int d1 = residual(b1, e4, 1, f[4], 1);   // [up, skip] :240
    int d2 = residual(d1, e3, 1, f[3], 1);
    int d3 = residual(d2, e2, 1, f[2], 1);
    int d4 = residual(d3, e1, 1, f[1], 1);
    b.logits(d4, -1);
}

}  // namespace

int build_graph(Engine& e) {
    switch (e.arch) {
        case PSEG_ARCH_FCN_SKIP: build_fcn(e, true); break;
        case PSEG_ARCH_FCN: build_fcn(e, false); break;
        case PSEG_ARCH_UNET: build_unet(e); break;
        case PSEG_ARCH_RES_UNET: build_res_unet(e); break;
        default: return fail(PSEG_EINVAL, "unknown architecture id %d", e.arch);
    }
    if ((e.flags & PSEG_FLAG_BATCHNORM) && e.arch != PSEG_ARCH_RES_UNET)
        return fail(PSEG_EUNSUPPORTED, "PSEG_FLAG_BATCHNORM: only the residual U-Net has BatchNormalization sites (lib/model.py:265-271)");
    // one timing slot per op
    for (auto& op : e.ops) {
        TimingSlot ts;
        ts.name = op.layer;
        e.slots.push_back(ts);
        op.timing_slot = (int)e.slots.size() - 1;
    }
    return PSEG_OK;
}

}  // namespace pseg
