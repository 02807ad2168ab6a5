// pseg_engine.hip -- engine object, f32-exact kernels, and the C ABI of include/pseg.h.
//
// F32_EXACT mode: every tensor is dense NHWC float32 with its true channel count.  Each output
// element is one sequential fmaf chain -- slabs of 16 input channels, (ky, kx, ci) inside a slab -- starting from +0, then "+ bias",
// then the optional residual add and ReLU -- the same operation sequence as
// oracle/pseg_oracle.c, so logits (and therefore label maps) are bit-identical to the oracle.
// This mode is the parity referee; the throughput mode lives in pseg_mfma.hip.
#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>

#include "pseg_common.h"

namespace pseg {

int build_graph(Engine& e);

std::string& last_error() {
    thread_local std::string msg;
    return msg;
}

// ---- knob snapshots (see pseg_common.h) -------------------------------------------------------------------------
struct KnobSnap {
    unsigned id = 0;
    std::map<std::string, std::string> kv;
};
namespace {
std::mutex g_knob_mu;
std::shared_ptr<const KnobSnap> g_knob_latest;     // what engine-less entries read
std::vector<std::shared_ptr<const KnobSnap>> g_knob_retired;
unsigned g_knob_next_id = 0;
thread_local const KnobSnap* tls_knobs = nullptr;  // the snapshot of the engine whose entry point is running on this thread
extern "C" char** environ;
}  // namespace
static const char* const k_env_knobs[] = {PSEG_ENV_KNOBS};
std::shared_ptr<const KnobSnap> knobs_snapshot(const char* plan) {
    auto m = std::make_shared<KnobSnap>();
#if PSEG_DIAG
    for (char** ep = environ; ep && *ep; ++ep) {          // (the diagnostic build answers from the live environment anyway)
        if (strncmp(*ep, "PSEG_", 5) != 0) continue;
        const char* eq = strchr(*ep, '=');
        if (eq) m->kv[std::string(*ep, eq - *ep)] = std::string(eq + 1);
    }
#else
    for (const char* name : k_env_knobs)
        if (const char* v = getenv(name)) m->kv[name] = v;
#endif
    // plan switches: "NAME=VALUE;NAME=VALUE" (pseg_create_plan); a name without '=' means "=1"
    for (const char* p = plan; p && *p;) {
        const char* e = strchr(p, ';');
        const std::string item = e ? std::string(p, e - p) : std::string(p);
        p = e ? e + 1 : p + item.size();
        if (item.compare(0, 5, "PSEG_") != 0) continue;
        const size_t eq = item.find('=');
        m->kv[item.substr(0, eq)] = eq == std::string::npos ? "1" : item.substr(eq + 1);
    }
    std::lock_guard<std::mutex> lk(g_knob_mu);
    // an unchanged environment (the normal case: one snapshot per process) shares the newest snapshot; a changed one
    // retires it -- retired "newest" snapshots stay alive (an engine-less call site on another thread may still hold a
    // pointer into it), one per CHANGE of the listed variables, not one per engine.  A snapshot with plan switches belongs to
    // its engine alone: it never becomes the newest one.
    if (g_knob_latest && g_knob_latest->kv == m->kv) return g_knob_latest;
    m->id = ++g_knob_next_id;
    if (plan && *plan) return m;
    if (g_knob_latest) g_knob_retired.push_back(g_knob_latest);
    g_knob_latest = m;
    return m;
}
static std::shared_ptr<const KnobSnap> knobs_latest() {
    {
        std::lock_guard<std::mutex> lk(g_knob_mu);
        if (g_knob_latest) return g_knob_latest;
    }
    return knobs_snapshot();
}
unsigned knob_generation() { return tls_knobs ? tls_knobs->id : knobs_latest()->id; }
const char* knob_lookup(const char* name) {
    // The returned pointer lives as long as the snapshot: an engine's snapshot for the engine's life; the newest one until
    // the next pseg_create -- a call site never hands out a cached pointer under another snapshot id (ids are never reused).
    const KnobSnap* k = tls_knobs;
    std::shared_ptr<const KnobSnap> hold;
    if (!k) { hold = knobs_latest(); k = hold.get(); }
    auto it = k->kv.find(name);
    return it == k->kv.end() ? nullptr : it->second.c_str();
}
KnobScope::KnobScope(const Engine& e) : prev(tls_knobs) { if (e.knobs) tls_knobs = e.knobs.get(); }
KnobScope::~KnobScope() { tls_knobs = prev; }

int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    last_error() = buf;
    return code;
}

// =============================================================================================
// f32-exact kernels
// =============================================================================================
constexpr int COT = 16;  // output channels per thread in the exact conv kernels


// One thread = one output pixel x COT consecutive output channels.  Weights are indexed only by
// loop counters and blockIdx, i.e. wave-uniform: hipcc serves them through the scalar cache.
// Chain order = the oracle's: slabs of PSEG_CHAIN_BLOCK input channels of the concatenated input, inside a slab (ky, kx, ci).
__global__ __launch_bounds__(256) void conv_exact_kernel(ConvArgs a) {
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= a.Hout * a.Wout) return;
    const int co0 = blockIdx.y * COT;
    const int y = pix / a.Wout, x = pix - y * a.Wout;
    const int Cin = a.C0 + a.C1;
    float acc[COT];
#pragma unroll
    for (int j = 0; j < COT; ++j) acc[j] = 0.0f;

    for (int cb = 0; cb < Cin; cb += PSEG_CHAIN_BLOCK) {
        const int ce = min(cb + PSEG_CHAIN_BLOCK, Cin);
        for (int ky = 0; ky < a.KH; ++ky) {
            const int iy = y * a.stride + ky - a.pt;
            if (iy < 0 || iy >= a.Hin) continue;
            for (int kx = 0; kx < a.KW; ++kx) {
                const int ix = x * a.stride + kx - a.pl;
                if (ix < 0 || ix >= a.Win) continue;
                const float* wt = a.w + (size_t)((ky * a.KW + kx) * Cin) * a.Cout + co0;
                const float* p0 = a.src0 + ((size_t)(iy >> a.up0) * (a.Win >> a.up0) + (ix >> a.up0)) * a.C0;
                const float* p1 = a.C1 > 0 ? a.src1 + ((size_t)(iy >> a.up1) * (a.Win >> a.up1) + (ix >> a.up1)) * a.C1 : nullptr;
                for (int ci = cb; ci < ce; ++ci) {
                    float xv;
                    if (ci < a.C0) {
                        xv = p0[ci];
                        if (a.mask) xv = a.mask[(p0 - a.src0) + ci] > 0.0f ? xv : 0.0f;
                    } else {
                        xv = p1[ci - a.C0];
                    }
                    if (a.in_relu) xv = xv > 0.0f ? xv : 0.0f;
                    const float* wr = wt + (size_t)ci * a.Cout;
#pragma unroll
                    for (int j = 0; j < COT; ++j) acc[j] = __builtin_fmaf(xv, wr[j], acc[j]);
                }
            }
        }
    }
    const size_t opix = a.dst_pitch ? (size_t)y * a.dst_pitch + x : (size_t)pix;
    float* o = a.dst + opix * a.Cout;
    const float* ad = a.add ? a.add + opix * a.Cout : nullptr;
#pragma unroll
    for (int j = 0; j < COT; ++j) {
        const int co = co0 + j;
        if (co < a.Cout) {
            float v = a.bias ? acc[j] + a.bias[co] : acc[j];
            if (ad) v = v + ad[co];
            if (a.relu) v = v > 0.0f ? v : 0.0f;
            o[co] = v;
        }
    }
}

// 1x1 convolution with a handful of output channels (the logits layer: 50 -> n_classes at full resolution).  In
// conv_exact_kernel a thread walks its pixel's channels in memory -- one cache line per lane and load; here a block takes
// 256 consecutive pixels of a row, whose channels are one contiguous run per source, through LDS with coalesced loads
// (pixel pitch Cin + 1 floats: conflict-free column reads), then runs the SAME chain per thread: ci ascending over
// [src0, src1], fmaf, + bias -- the same bits.  HBM-bound (Cin * 4 B read per pixel): 0.83 ms -> see DESIGN 5.
__global__ __launch_bounds__(256) void conv1x1_exact_kernel(ConvArgs a) {
    extern __shared__ float xs[];   // [256][Cin + 1]
    const int Cin = a.C0 + a.C1, P = Cin + 1;
    const int y = blockIdx.y, x0 = blockIdx.x * 256;
    const int npx = min(256, a.Wout - x0);
    const int co0 = blockIdx.z * COT;
    for (int srcsel = 0; srcsel < (a.C1 > 0 ? 2 : 1); ++srcsel) {
        const int C = srcsel ? a.C1 : a.C0, cbase = srcsel ? a.C0 : 0;
        const float* p = (srcsel ? a.src1 : a.src0) + ((size_t)y * a.Win + x0) * C;
        const int n = npx * C;
        // e / C by a reciprocal multiply: floor(e * inv / 2^32) with inv = floor(2^32 / C) + 1 is exact while e * C < 2^32
        // (here e < 256 * C <= 2^15).  (A 20-bit reciprocal, exact only while e * C < 2^20, was wrong for 73 <= C <= 127.)
        const unsigned long long inv = (1ull << 32) / (unsigned)C + 1ull;
        for (int e0 = 0; e0 < n; e0 += 256 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + u * 256 + (int)threadIdx.x;
                v[u] = e < n ? p[e] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + u * 256 + (int)threadIdx.x;
                const int px = (int)(((unsigned long long)(unsigned)e * inv) >> 32), c = e - px * C;
                if (e < n) xs[px * P + cbase + c] = v[u];
            }
        }
    }
    __syncthreads();
    const int t = threadIdx.x;
    if (t >= npx) return;
    float acc[COT];
#pragma unroll
    for (int j = 0; j < COT; ++j) acc[j] = 0.0f;
    const float* wt = a.w + co0;
    const float* xp = xs + t * P;
    for (int ci = 0; ci < Cin; ++ci) {
        const float xv = xp[ci];
        const float* wr = wt + (size_t)ci * a.Cout;
#pragma unroll
        for (int j = 0; j < COT; ++j) acc[j] = __builtin_fmaf(xv, wr[j], acc[j]);
    }
    const int x = x0 + t;
    const size_t opix = a.dst_pitch ? (size_t)y * a.dst_pitch + x : (size_t)y * a.Wout + x;
    float* o = a.dst + opix * a.Cout;
    const float* ad = a.add ? a.add + opix * a.Cout : nullptr;
#pragma unroll
    for (int j = 0; j < COT; ++j) {
        const int co = co0 + j;
        if (co < a.Cout) {
            float v = a.bias ? acc[j] + a.bias[co] : acc[j];
            if (ad) v = v + ad[co];
            if (a.relu) v = v > 0.0f ? v : 0.0f;
            o[co] = v;
        }
    }
}

struct DeconvArgs {
    const float* src0;
    const float* src1;
    int C0, C1;
    int Hin, Win;
    const float* w;  // [2][2][Cin][Cout]
    const float* bias;
    float* dst;      // (2Hin) x (2Win) x Cout
    int Cout, relu;
};

// blockIdx.z = a*2+b keeps the tap (and with it the weight row) wave-uniform.
__global__ __launch_bounds__(256) void deconv2_exact_kernel(DeconvArgs a) {
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= a.Hin * a.Win) return;
    const int co0 = blockIdx.y * COT;
    const int ab = blockIdx.z;
    const int i = pix / a.Win, j0 = pix - i * a.Win;
    const int Cin = a.C0 + a.C1;
    float acc[COT];
#pragma unroll
    for (int j = 0; j < COT; ++j) acc[j] = 0.0f;
    const float* wt = a.w + (size_t)ab * Cin * a.Cout + co0;
    const float* p0 = a.src0 + (size_t)pix * a.C0;
    for (int ci = 0; ci < a.C0; ++ci) {
        const float xv = p0[ci];
        const float* wr = wt + (size_t)ci * a.Cout;
#pragma unroll
        for (int j = 0; j < COT; ++j) acc[j] = __builtin_fmaf(xv, wr[j], acc[j]);
    }
    if (a.C1 > 0) {
        const float* p1 = a.src1 + (size_t)pix * a.C1;
        const float* wt1 = wt + (size_t)a.C0 * a.Cout;
        for (int ci = 0; ci < a.C1; ++ci) {
            const float xv = p1[ci];
            const float* wr = wt1 + (size_t)ci * a.Cout;
#pragma unroll
            for (int j = 0; j < COT; ++j) acc[j] = __builtin_fmaf(xv, wr[j], acc[j]);
        }
    }
    const int oy = 2 * i + (ab >> 1), ox = 2 * j0 + (ab & 1);
    float* o = a.dst + ((size_t)oy * (2 * a.Win) + ox) * a.Cout;
#pragma unroll
    for (int j = 0; j < COT; ++j) {
        const int co = co0 + j;
        if (co < a.Cout) {
            float v = acc[j] + a.bias[co];
            if (a.relu) v = v > 0.0f ? v : 0.0f;
            o[co] = v;
        }
    }
}

__global__ void pool_exact_kernel(const float* in, int H, int W, int C, float* out) {
    const size_t n = (size_t)(H / 2) * (W / 2) * C;
    const int Wo = W / 2;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n;
         t += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(t % C);
        const size_t p = t / C;
        const int x = (int)(p % Wo), y = (int)(p / Wo);
        const float* b = in + ((size_t)(2 * y) * W + 2 * x) * C + c;
        const float v0 = b[0], v1 = b[C], v2 = b[(size_t)W * C], v3 = b[(size_t)W * C + C];
        float m = v0 > v1 ? v0 : v1;
        const float n2 = v2 > v3 ? v2 : v3;
        out[t] = m > n2 ? m : n2;
    }
}

// x/255 via the host-built LUT (bit-identical to the oracle's float division), zero pad to the
// 32-multiple canvas (lib/model.py:20-26).
__global__ void preprocess_exact_kernel(const uint8_t* img, int H, int W, int C, const float* lut,
                                        float* dst, int Hp, int Wp) {
    const size_t n = (size_t)Hp * Wp * C;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n;
         t += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(t % C);
        const size_t p = t / C;
        const int x = (int)(p % Wp), y = (int)(p / Wp);
        dst[t] = (y < H && x < W) ? lut[img[((size_t)y * W + x) * C + c]] : 0.0f;
    }
}

// float page (augmented training samples, values on the 0..255 scale): preprocess = x / 255.0f (numpy float32 / python float)
__global__ void preprocess_exact_f32_kernel(const float* img, int H, int W, int C, float* dst, int Hp, int Wp) {
    const size_t n = (size_t)Hp * Wp * C;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n;
         t += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(t % C);
        const size_t p = t / C;
        const int x = (int)(p % Wp), y = (int)(p / Wp);
        dst[t] = (y < H && x < W) ? img[((size_t)y * W + x) * C + c] / 255.0f : 0.0f;
    }
}

// softmax(-1) and argmax(-1) of the float32 logits (lib/network.py:258-259): max-subtracted
// exp / sum in f32 (scipy.special.softmax on f32 input); argmax first-maximum-wins.
__global__ void softmax_argmax_kernel(const float* logits, size_t n, int C, float* probs,
                                      int64_t* labels, uint8_t* labels_u8) {
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n;
         p += (size_t)gridDim.x * blockDim.x) {
        const float* z = logits + p * C;
        int best = 0;
        float bv = z[0];
        for (int c = 1; c < C; ++c) {
            const float v = z[c];
            if (v > bv) { bv = v; best = c; }
        }
        if (labels) labels[p] = best;
        if (labels_u8) labels_u8[p] = (uint8_t)best;
        if (probs) {
            float s = 0.0f;
            for (int c = 0; c < C; ++c) s += expf(z[c] - bv);
            for (int c = 0; c < C; ++c) probs[p * C + c] = expf(z[c] - bv) / s;
        }
    }
}

// top-1 minus top-2 logit per pixel (the quantity the label-exact mode thresholds); C == 1 gives +inf
__global__ void margin_from_logits_kernel(const float* logits, size_t n, int C, float* margin) {
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
        const float* z = logits + p * C;
        float bv = z[0], sv = -3.4e38f;
        for (int c = 1; c < C; ++c) {
            const float v = z[c];
            if (v > bv) { sv = bv; bv = v; }
            else if (v > sv) sv = v;
        }
        margin[p] = C > 1 ? bv - sv : __builtin_inff();
    }
}
void launch_margin_from_logits(const float* d_logits, size_t n, int C, float* d_margin, hipStream_t st) {
    margin_from_logits_kernel<<<(int)std::min<size_t>((n + 255) / 256, 8192), 256, 0, st>>>(d_logits, n, C, d_margin);
}

// =============================================================================================
// timing helpers (HIP events on the launch stream)
// =============================================================================================
static int get_event(Engine& e, hipEvent_t* ev) {
    if (!e.event_pool.empty()) {
        *ev = e.event_pool.back();
        e.event_pool.pop_back();
        return PSEG_OK;
    }
    PSEG_HIP(hipEventCreate(ev));
    return PSEG_OK;
}

int time_begin(Engine& e, Op& op, hipStream_t st, hipEvent_t* ev0) {
    *ev0 = nullptr;
    if (!e.timing || op.timing_slot < 0) return PSEG_OK;
    PSEG_TRY(get_event(e, ev0));
    PSEG_HIP(hipEventRecord(*ev0, st));
    return PSEG_OK;
}

int time_end(Engine& e, Op& op, hipStream_t st, hipEvent_t ev0) {
    if (!ev0) return PSEG_OK;
    hipEvent_t ev1;
    PSEG_TRY(get_event(e, &ev1));
    PSEG_HIP(hipEventRecord(ev1, st));
    e.slots[op.timing_slot].pending.emplace_back(ev0, ev1);
    return PSEG_OK;
}

static int timing_collect(Engine& e) {
    for (auto& s : e.slots) {
        for (auto& pr : s.pending) {
            PSEG_HIP(hipEventSynchronize(pr.second));
            float ms = 0;
            PSEG_HIP(hipEventElapsedTime(&ms, pr.first, pr.second));
            s.total_ms += ms;
            s.launches += 1;
            e.event_pool.push_back(pr.first);
            e.event_pool.push_back(pr.second);
        }
        s.pending.clear();
    }
    return PSEG_OK;
}

// =============================================================================================
// engine: weights, canvas, run
// =============================================================================================
static void free_dev(void*& p) {
    if (p) (void)hipFree(p);
    p = nullptr;
}

static int ensure(void** p, size_t* cap, size_t bytes) {
    if (*cap >= bytes && *p) return PSEG_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    PSEG_HIP(hipMalloc(p, bytes));
    *cap = bytes;
    return PSEG_OK;
}

// Upload weights in correlation form.  Conv2D: as is.  Conv2DTranspose s1 (kh,kw,Cout,Cin):
// Wc[ky][kx][ci][co] = K[KH-1-ky][KW-1-kx][co][ci].  Conv2DTranspose k2 s2: [a][b][ci][co] =
// K[a][b][co][ci].
int launch_conv_exact(const ConvArgs& a, hipStream_t st, bool* pooled) {
    // matrix-core path (same bits, see pseg_exact_mfma.hip)
    if (pooled) *pooled = false;
    const int rv = a.pool_dst ? 0 : launch_conv_first_valu(a, st);     // 1 or 3 input channels: HBM-bound, vector ALU
    if (rv != 0) return rv < 0 ? rv : PSEG_OK;
    const int rc = launch_conv_exact_mfma(a, st);
    if (rc != 0) {
        if (pooled) *pooled = rc == 2;
        return rc < 0 ? rc : PSEG_OK;
    }
    const int Cin = a.C0 + a.C1;
    if (a.KH == 1 && a.KW == 1 && a.stride == 1 && !a.pt && !a.pl && !a.up0 && !a.up1 && !a.in_relu && !a.mask && Cin <= 127) {
        static bool attr_set[64] = {false};
        int dev = 0;
        PSEG_HIP(hipGetDevice(&dev));
        if (!attr_set[dev & 63]) {
            PSEG_HIP(hipFuncSetAttribute((const void*)conv1x1_exact_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_set[dev & 63] = true;
        }
        dim3 g1(cdiv(a.Wout, 256), a.Hout, cdiv(a.Cout, COT));
        conv1x1_exact_kernel<<<g1, 256, (size_t)256 * (Cin + 1) * sizeof(float), st>>>(a);
        PSEG_HIP(hipGetLastError());
        return PSEG_OK;
    }
    dim3 grid(cdiv(a.Hout * a.Wout, 256), cdiv(a.Cout, COT));
    conv_exact_kernel<<<grid, 256, 0, st>>>(a);
    PSEG_HIP(hipGetLastError());
    return PSEG_OK;
}

// OP_BN: this op's channel slice of the layer's four vectors; bf16 mode: folded to per-channel scale / shift over the
// storage channels (zero on the pad channels, so they stay zero)
static int upload_bn(Engine& e, Op& op) {
    const int C = op.Cin, c0 = op.bn_c0;
    const int pidx[4] = {op.kparam, op.bparam, op.mmparam, op.mvparam};
    std::vector<float> par((size_t)4 * C);
    for (int j = 0; j < 4; ++j)
        for (int c = 0; c < C; ++c) par[(size_t)j * C + c] = e.params[pidx[j]].host[(size_t)c0 + c];
    free_dev((void*&)op.d_w);
    free_dev((void*&)op.d_b);
    if (e.mode == PSEG_MODE_BF16) {
        const int Cs = e.tensors[op.src0].Cs;
        std::vector<float> ss((size_t)2 * Cs, 0.0f);
        for (int c = 0; c < C; ++c) {
            const float scale = par[c] / std::sqrt(par[(size_t)3 * C + c] + PSEG_BN_EPS);
            ss[c] = scale;
            ss[(size_t)Cs + c] = par[(size_t)C + c] - par[(size_t)2 * C + c] * scale;
        }
        PSEG_HIP(hipMalloc((void**)&op.d_w, ss.size() * sizeof(float)));
        PSEG_HIP(hipMemcpy(op.d_w, ss.data(), ss.size() * sizeof(float), hipMemcpyHostToDevice));
        return PSEG_OK;
    }
    PSEG_HIP(hipMalloc((void**)&op.d_w, par.size() * sizeof(float)));
    PSEG_HIP(hipMemcpy(op.d_w, par.data(), par.size() * sizeof(float), hipMemcpyHostToDevice));
    PSEG_HIP(hipMalloc((void**)&op.d_b, bn_saved_bytes(C)));
    PSEG_HIP(hipMemset(op.d_b, 0, bn_saved_bytes(C)));
    return PSEG_OK;
}

int upload_weights(Engine& e) {
    // the copies below are ordered on the null stream only: earlier asynchronous predicts must have finished
    PSEG_HIP(hipDeviceSynchronize());
    for (auto& op : e.ops) {
        if (op.kparam < 0) continue;
        if (op.type == OP_BN) {
            PSEG_TRY(upload_bn(e, op));
            continue;
        }
        const Param& kp = e.params[op.kparam];
        const Param& bp = e.params[op.bparam];
        const int k = op.k, Cin = op.Cin, Cout = op.Cout;
        std::vector<float> w((size_t)k * k * Cin * Cout + COT, 0.0f);
        for (int ky = 0; ky < k; ++ky)
            for (int kx = 0; kx < k; ++kx)
                for (int ci = 0; ci < Cin; ++ci)
                    for (int co = 0; co < Cout; ++co) {
                        float v;
                        if (!op.transposed)
                            v = kp.host[(((size_t)ky * k + kx) * Cin + ci) * Cout + co];
                        else if (op.type == OP_DECONV2)
                            v = kp.host[(((size_t)ky * k + kx) * Cout + co) * Cin + ci];
                        else
                            v = kp.host[(((size_t)(k - 1 - ky) * k + (k - 1 - kx)) * Cout + co) * Cin + ci];
                        w[(((size_t)ky * k + kx) * Cin + ci) * Cout + co] = v;
                    }
        free_dev((void*&)op.d_w);
        free_dev((void*&)op.d_b);
        PSEG_HIP(hipMalloc((void**)&op.d_w, w.size() * sizeof(float)));
        PSEG_HIP(hipMemcpy(op.d_w, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice));
        PSEG_HIP(hipMalloc((void**)&op.d_b, (size_t)round_up(Cout, COT) * sizeof(float)));
        PSEG_HIP(hipMemset(op.d_b, 0, (size_t)round_up(Cout, COT) * sizeof(float)));
        PSEG_HIP(hipMemcpy(op.d_b, bp.host.data(), (size_t)Cout * sizeof(float), hipMemcpyHostToDevice));
        // float32 engine: room for the left-over output channels' shifted kernel copies (conv_xb_kernel REM), rebuilt by the
        // launcher on the first launch after every weight change
        op.wrem_valid = false;
        const size_t wrb = (e.mode != PSEG_MODE_BF16 && op.type == OP_CONV && op.stride == 1) ? wrem_bytes_for(k, k, Cin, Cout) : 0;
        if (wrb != op.wrem_bytes) {
            free_dev((void*&)op.d_wrem);
            op.wrem_bytes = 0;
            if (wrb) { PSEG_HIP(hipMalloc((void**)&op.d_wrem, wrb)); op.wrem_bytes = wrb; }
        }
        if (e.mode == PSEG_MODE_BF16) PSEG_TRY(mfma_pack_op(e, op, w, bp.host));
    }
    // copies from pageable memory may still be in their final DMA when hipMemcpy returns, and the null stream is
    // not ordered with the engine's non-blocking stream: finish them before any kernel can read the weights
    PSEG_HIP(hipDeviceSynchronize());
    e.weights_dirty = false;
    return PSEG_OK;
}

int set_canvas(Engine& e, int H, int W, hipStream_t st, int pages) {
    if (H <= 0 || W <= 0) return fail(PSEG_EINVAL, "empty page %dx%d", H, W);
    const int Hp = round_up(H, 32), Wp = round_up(W, 32);
    e.H = H;
    e.W = W;
    if (Hp == e.Hp && Wp == e.Wp && pages <= e.pages) return PSEG_OK;
    pages = std::max(pages, Hp == e.Hp && Wp == e.Wp ? e.pages : 1);
    // A canvas change re-allocates and clears the activation tensors.  Work of earlier calls may still be in
    // flight on the engine's (non-blocking) stream or on a caller's stream: drain the device first, and clear on
    // the stream the coming kernels run on -- a hipMemset on the null stream is NOT ordered with non-blocking
    // streams (that race corrupted the first predict after shrinking from a 4096x3072 canvas).
    const size_t esz = e.mode == PSEG_MODE_BF16 ? 2 : 4;
    if (e.mode == PSEG_MODE_BF16) {
        // the throughput kernels address every tensor through 32-bit buffer descriptors / offsets (out-of-range reads give
        // zeros, out-of-range stores are dropped -- silently wrong labels, not a fault): refuse canvases whose largest
        // tensor, or the 16 B/px skip-logits / margin planes, reach 4 GiB (unet: 64 bf16 channels -> about 5790 x 5790)
        // (per page slot: every launch, also one over several slots, builds its descriptors on the slot's own base)
        size_t worst = (size_t)Hp * Wp * 16 * 4;
        for (auto& t : e.tensors) worst = std::max(worst, (size_t)(Hp >> t.s) * (Wp >> t.s) * t.Cs * esz);
        if (worst >= ((size_t)1 << 32))
            return fail(PSEG_EUNSUPPORTED, "page %dx%d: a tensor of this graph would reach 4 GiB (32-bit buffer addressing in the bf16 kernels); "
                                           "predict it in tiles or use the float32 mode", H, W);
    }
    PSEG_HIP(hipDeviceSynchronize());
    // The new canvas is committed only when every tensor has its buffer: a failed allocation leaves the engine with NO canvas
    // (Hp = Wp = 0, one slot, every pointer null or valid for its recorded size), so the next call allocates again instead of
    // returning early on the "same shape" test and running kernels on freed or undersized buffers.
    e.Hp = e.Wp = 0;
    e.pages = 1;
    for (auto& t : e.tensors) {
        const size_t pb = (size_t)(Hp >> t.s) * (Wp >> t.s) * t.Cs * esz;
        const size_t bytes = pb * pages;
        if (bytes > t.bytes) {
            free_dev(t.base);
            t.d = nullptr;
            t.bytes = 0;
            const hipError_t er = hipMalloc(&t.base, bytes);
            if (er != hipSuccess) {
                (void)hipGetLastError();
                t.base = nullptr;
                return fail(er == hipErrorOutOfMemory ? PSEG_ENOMEM : PSEG_EHIP, "canvas %dx%d x %d page slot(s): hipMalloc of %zu bytes for tensor '%s' failed: %s",
                            Hp, Wp, pages, bytes, t.name.c_str(), hipGetErrorString(er));
            }
            t.bytes = bytes;
        }
        t.page_bytes = pb;
        t.d = t.base;
        // float32 mode: fresh buffers start at zero; bf16 mode: the pad channels no kernel writes must read as
        // zero after every layout change
        if (t.d && bytes) PSEG_HIP(hipMemsetAsync(t.d, 0, bytes, st));
    }
    e.Hp = Hp;
    e.Wp = Wp;
    e.pages = pages;
    const double px = (double)Hp * Wp;
    for (auto& op : e.ops) e.slots[op.timing_slot].flops = op.flops_per_canvas_px * px;
    // a launch that also runs a fused-away layer (conv1 inside conv2, logits inside the tail) does that layer's work too
    for (auto& op : e.ops) {
        if (op.fuse1 >= 0) e.slots[op.timing_slot].flops += e.ops[op.fuse1].flops_per_canvas_px * px;
        if (op.tail_logits >= 0) e.slots[op.timing_slot].flops += e.ops[op.tail_logits].flops_per_canvas_px * px;
        if (op.into_tail >= 0) e.slots[e.ops[op.into_tail].timing_slot].flops += op.flops_per_canvas_px * px;
    }
    return PSEG_OK;
}

// inverted dropout with a counter-based mask: element i is kept iff the top 24 bits of a 32-bit mix of (i, key)
// are >= rate * 2^24; the same call on a gradient tensor applies the same mask and scale (tests/test_train_arch_gpu.py
// restates the mix in NumPy)
__global__ void dropout_kernel(float* x, size_t n, uint32_t key, uint32_t thresh, float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 0x9E3779B1u + key;
        h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
        x[i] = (h >> 8) >= thresh ? x[i] * scale : 0.0f;
    }
}
void launch_dropout(float* x, size_t n, uint32_t key, float rate, hipStream_t st) {
    const uint32_t thresh = (uint32_t)(rate * 16777216.0f);
    dropout_kernel<<<(int)std::min<size_t>((n + 255) / 256, 8192), 256, 0, st>>>(x, n, key, thresh, 1.0f / (1.0f - rate));
}

int engine_status(Engine& e, hipStream_t st) {
    if (!e.d_sp_err) {                    // (float32 engines, graphs without a conv_sp_kernel layer)
        PSEG_HIP(hipStreamSynchronize(st));
        return PSEG_OK;
    }
    PSEG_HIP(hipMemcpyAsync(e.h_sp_err, e.d_sp_err, 32, hipMemcpyDeviceToHost, st));
    PSEG_HIP(hipStreamSynchronize(st));
    if (e.h_sp_err[0] == 0) return PSEG_OK;
    int r[8];
    memcpy(r, e.h_sp_err, 32);
    PSEG_HIP(hipMemsetAsync(e.d_sp_err, 0, 32, st));          // reported once: the next call starts clean
    PSEG_HIP(hipStreamSynchronize(st));
    const char* layer = r[7] >= 0 && r[7] < (int)e.ops.size() ? e.ops[r[7]].layer.c_str() : "?";
    return fail(PSEG_EHIP, "conv_sp_kernel (%s): a counter wait gave up -- the results since the last status check are not valid "
                           "(code %d: 1 tile loader, 2 weight loader, 3 compute; wave %d of workgroup %d needed %d had %d / needed %d had %d)",
                layer, r[0], r[1], r[6], r[2], r[3], r[4], r[5]);
}

// Page slots that fit the device: `want` slots of this canvas, halved until the activation tensors of the unit (what the engine
// already holds counts as free) leave a fifth of the free memory untouched.  >= 1 (one slot is what a single page needs anyway).
static int fit_page_slots(Engine& e, int H, int W, int want) {
    if (want <= 1) return 1;
    const int Hp = round_up(H, 32), Wp = round_up(W, 32);
    const size_t esz = e.mode == PSEG_MODE_BF16 ? 2 : 4;
    size_t per_slot = (size_t)Hp * Wp * 16, held = 0;          // (skip logits: 16 B/px per slot)
    for (auto& t : e.tensors) { per_slot += (size_t)(Hp >> t.s) * (Wp >> t.s) * t.Cs * esz; held += t.bytes; }
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); return want; }
    const double room = 0.8 * ((double)fr + (double)held);
    while (want > 1 && (double)per_slot * want > room) want = (want + 1) / 2;
    return want;
}

int run_exact(Engine& e, const uint8_t* d_img, float* d_logits, float* d_probs,
                     int64_t* d_labels, uint8_t* d_labels_u8, hipStream_t st) {
    Tensor& in = e.tensors[e.input_tensor];
    {
        const size_t n = (size_t)e.Hp * e.Wp * in.C;
        const int grid = (int)std::min<size_t>((n + 255) / 256, 4096);
        if (e.cur_img_f32)
            preprocess_exact_f32_kernel<<<grid, 256, 0, st>>>(e.cur_img_f32, e.H, e.W, in.C, (float*)in.d, e.Hp, e.Wp);
        else
            preprocess_exact_kernel<<<grid, 256, 0, st>>>(d_img, e.H, e.W, in.C, e.d_lut, (float*)in.d,
                                                          e.Hp, e.Wp);
    }
    const Op* skip_pool = nullptr;
    const Op* fused_logits = nullptr;
    for (auto& op : e.ops) {
        hipEvent_t ev0;
        PSEG_TRY(time_begin(e, op, st, &ev0));
        const Tensor& s0 = e.tensors[op.src0];
        const Tensor* s1 = op.src1 >= 0 ? &e.tensors[op.src1] : nullptr;
        if (&op == fused_logits) {
            // logits + argmax ran inside the transposed conv in front of it
        } else if (op.type == OP_CONV || op.type == OP_LOGITS) {
            ConvArgs a{};
            a.src0 = (const float*)s0.d;
            a.src1 = s1 ? (const float*)s1->d : nullptr;
            a.C0 = s0.C;
            a.C1 = s1 ? s1->C : 0;
            a.up0 = op.up0;
            a.up1 = op.up1;
            a.Hin = e.tH(s0) << op.up0;
            a.Win = e.tW(s0) << op.up0;
            a.w = op.d_w;
            a.bias = op.d_b;
            a.KH = a.KW = op.k;
            a.stride = op.stride;
            a.in_relu = op.in_relu;
            a.relu = op.relu;
            a.Cout = op.Cout;
            a.relaxed = e.relaxed_f32;
            a.wrem_buf = op.d_wrem; a.wrem_cap = op.wrem_bytes; a.wrem_valid = &op.wrem_valid;
            if (op.type == OP_LOGITS) {
                // crop (lib/model.py:29-42) folded into the output extent
                a.Hout = e.H;
                a.Wout = e.W;
                a.pt = a.pl = 0;
                float* zl = d_logits;
                if (!zl) {
                    PSEG_TRY(ensure((void**)&e.d_logits_tmp, &e.logits_tmp_bytes,
                                    (size_t)e.H * e.W * op.Cout * sizeof(float)));
                    zl = e.d_logits_tmp;
                }
                a.dst = zl;
                // the kernel indexes src rows with Win: rows of the padded canvas
                // logical input dims stay the canvas; output (y,x) reads input (y,x)
                PSEG_TRY(launch_conv_exact(a, st));
                if (d_probs || d_labels || d_labels_u8) {
                    const size_t n = (size_t)e.H * e.W;
                    const int g2 = (int)std::min<size_t>((n + 255) / 256, 8192);
                    softmax_argmax_kernel<<<g2, 256, 0, st>>>(zl, n, op.Cout, d_probs, d_labels,
                                                             d_labels_u8);
                }
            } else {
                const Tensor& d = e.tensors[op.dst];
                a.Hout = e.tH(d);
                a.Wout = e.tW(d);
                // TF SAME: pad_total = max((out-1)*s + k - in, 0); before = total / 2
                const int tot_h = std::max((a.Hout - 1) * op.stride + op.k - a.Hin, 0);
                const int tot_w = std::max((a.Wout - 1) * op.stride + op.k - a.Win, 0);
                a.pt = tot_h / 2;
                a.pl = tot_w / 2;
                if (op.transposed) {  // flipped kernel: "before" and "after" swap (odd k: equal)
                    a.pt = tot_h - a.pt;
                    a.pl = tot_w - a.pl;
                }
                a.add = op.add >= 0 ? (const float*)e.tensors[op.add].d : nullptr;
                a.dst = (float*)d.d;
                // MaxPooling2D right behind this layer (every encoder stage of fcn / unet): pooled in the conv's epilogue
                const Op* nx = (&op + 1 < e.ops.data() + e.ops.size()) ? &op + 1 : nullptr;
                // (measured, fcn_skip 2048x1536: the fused form saves the pool passes -- 91 + 27 + 13 us -- and costs the three
                // producers 84 + 40 + 12 us in their epilogues: no gain, so the separate streaming pass stays the default;
                // PSEG_EXACT_POOL_FUSE=1 selects the fused form, tests keep both bit-identical)
                const bool want_pool = nx && nx->type == OP_POOL && nx->src0 == op.dst && op.add < 0 && !(e.drop_key && op.dropout > 0.0f) &&
                                       PSEG_KNOB("PSEG_EXACT_POOL_FUSE");
                if (want_pool) a.pool_dst = (float*)e.tensors[nx->dst].d;
                bool pooled = false;
                PSEG_TRY(launch_conv_exact(a, st, &pooled));
                skip_pool = pooled ? nx : nullptr;
            }
        } else if (op.type == OP_BN) {
            const size_t npx = (size_t)e.tH(s0) * e.tW(s0);
            float* y = (float*)e.tensors[op.dst].d;
            if (e.bn_training) PSEG_TRY(bn_train_forward((const float*)s0.d, y, npx, op.Cin, op.d_w, op.d_b, op.relu, op.up0, st));
            else PSEG_TRY(bn_infer((const float*)s0.d, y, npx, op.Cin, op.d_w, op.d_b, op.relu, st));
        } else if (op.type == OP_DECONV2) {
            // Conv2DTranspose k2 s2: HBM-bound -- one output pixel per thread on the vector ALU (pseg_exact_valu.hip); when
            // its only reader is the logits layer (fcn_skip / fcn: deconv5), that layer and the argmax run in the same
            // kernel on the values still in registers.  Fallbacks: the matrix-core GEMM form, then the scalar kernel.
            bool done = false;
            {
                TailArgs ta{};
                ta.src0 = (const float*)s0.d; ta.src1 = s1 ? (const float*)s1->d : nullptr;
                ta.C0 = s0.C; ta.C1 = s1 ? s1->C : 0; ta.Hin = e.tH(s0); ta.Win = e.tW(s0);
                ta.w = op.d_w; ta.bias = op.d_b; ta.Cout = op.Cout; ta.relu = op.relu;
                ta.dst = (float*)e.tensors[op.dst].d;
                const Op* nx = (&op + 1 < e.ops.data() + e.ops.size()) ? &op + 1 : nullptr;
                bool tail = nx && nx->type == OP_LOGITS && nx->src0 == op.dst && nx->k == 1 && !op.relu && op.Cout == 20 && nx->Cout <= 8 &&
                            !(e.drop_key && op.dropout > 0.0f) && !PSEG_KNOB("PSEG_EXACT_NO_TAIL_FUSE");
                if (tail) {
                    const Tensor* sk = nx->src1 >= 0 ? &e.tensors[nx->src1] : nullptr;
                    if (sk && sk->s != e.tensors[op.dst].s) tail = false;
                    if (tail) {
                        ta.skip = sk ? (const float*)sk->d : nullptr; ta.Cs = sk ? sk->C : 0;
                        ta.wl = nx->d_w; ta.bl = nx->d_b; ta.ncls = nx->Cout; ta.H = e.H; ta.W = e.W;
                        float* zl = d_logits;
                        if (!zl && d_probs) {
                            PSEG_TRY(ensure((void**)&e.d_logits_tmp, &e.logits_tmp_bytes, (size_t)e.H * e.W * nx->Cout * sizeof(float)));
                            zl = e.d_logits_tmp;
                        }
                        ta.logits = zl; ta.labels = d_labels; ta.labels_u8 = d_labels_u8;
                    }
                }
                int rv = launch_deconv2_valu(ta, tail, st);
                if (rv == 0 && tail) { tail = false; rv = launch_deconv2_valu(ta, false, st); }
                if (rv < 0) return rv;
                done = rv == 1;
                if (done && tail) {
                    fused_logits = nx;
                    if (d_probs) {
                        const size_t n = (size_t)e.H * e.W;
                        softmax_argmax_kernel<<<(int)std::min<size_t>((n + 255) / 256, 8192), 256, 0, st>>>(ta.logits, n, nx->Cout, d_probs, nullptr, nullptr);
                    }
                }
            }
            if (!done) {
                ConvArgs c{};
                c.src0 = (const float*)s0.d;
                c.src1 = s1 ? (const float*)s1->d : nullptr;
                c.C0 = s0.C;
                c.C1 = s1 ? s1->C : 0;
                c.Hin = c.Hout = e.tH(s0);
                c.Win = c.Wout = e.tW(s0);
                c.w = op.d_w;
                c.bias = op.d_b;
                c.KH = c.KW = 1;
                c.stride = 1;
                c.Cout = op.Cout;
                c.relu = op.relu;
                c.dst = (float*)e.tensors[op.dst].d;
                c.dst_pitch = 2 * c.Win;
                c.deconv4 = 1;
                c.relaxed = e.relaxed_f32;
                const int rc = launch_conv_exact_mfma(c, st);
                if (rc < 0) return rc;
                done = rc == 1;
            }
            if (!done) {
            DeconvArgs a{};
            a.src0 = (const float*)s0.d;
            a.src1 = s1 ? (const float*)s1->d : nullptr;
            a.C0 = s0.C;
            a.C1 = s1 ? s1->C : 0;
            a.Hin = e.tH(s0);
            a.Win = e.tW(s0);
            a.w = op.d_w;
            a.bias = op.d_b;
            a.dst = (float*)e.tensors[op.dst].d;
            a.Cout = op.Cout;
            a.relu = op.relu;
            dim3 grid(cdiv(a.Hin * a.Win, 256), cdiv(op.Cout, COT), 4);
            deconv2_exact_kernel<<<grid, 256, 0, st>>>(a);
            }
        } else if (op.type == OP_POOL && &op == skip_pool) {
            // done in the producing conv's epilogue
        } else if (op.type == OP_POOL) {
            const size_t n = (size_t)(e.tH(s0) / 2) * (e.tW(s0) / 2) * s0.C;
            const int grid = (int)std::min<size_t>((n + 255) / 256, 8192);
            pool_exact_kernel<<<grid, 256, 0, st>>>((const float*)s0.d, e.tH(s0), e.tW(s0), s0.C,
                                                    (float*)e.tensors[op.dst].d);
        }
        if (e.drop_key && op.dropout > 0.0f) {   // training forward: Dropout on this op's output, in place
            const Tensor& d = e.tensors[op.dst];
            launch_dropout((float*)d.d, (size_t)e.tH(d) * e.tW(d) * d.C, e.drop_key + 0x85EBCA77u * (uint32_t)(&op - e.ops.data()), op.dropout, st);
        }
        PSEG_HIP(hipGetLastError());
        PSEG_TRY(time_end(e, op, st, ev0));
    }
    return PSEG_OK;
}

static int run_bf16(Engine& e, const uint8_t* d_img, float* d_logits, float* d_probs,
                    int64_t* d_labels, uint8_t* d_labels_u8, float* d_margin, hipStream_t st) {
    PSEG_TRY(mfma_preprocess(e, d_img, st));
    e.cur_margin = d_margin;
    e.margin_done = false;
    if (d_margin && !d_logits && !mfma_tail_emits_margin(e)) {
        // this graph's tail kernel has no margin output: take the float32 logits and derive the margin from them
        PSEG_TRY(ensure((void**)&e.d_logits_tmp, &e.logits_tmp_bytes, (size_t)e.H * e.W * e.n_classes * sizeof(float)));
        d_logits = e.d_logits_tmp;
    }
    e.cur_logits = d_logits;
    e.cur_probs = d_probs;
    e.cur_labels = d_labels;
    e.cur_labels_u8 = d_labels_u8;
    for (auto& op : e.ops) {
        if (op.fused_away) continue;
        hipEvent_t ev0;
        PSEG_TRY(time_begin(e, op, st, &ev0));
        switch (op.type) {
            case OP_CONV: PSEG_TRY(mfma_launch_conv(e, op, st)); break;
            case OP_DECONV2: PSEG_TRY(mfma_launch_deconv2(e, op, st)); break;
            case OP_POOL: PSEG_TRY(mfma_launch_pool(e, op, st)); break;
            case OP_BN: {
                const Tensor& s0 = e.tensors[op.src0];
                PSEG_TRY(bn_infer_bf16((const uint16_t*)s0.d, (uint16_t*)e.tensors[op.dst].d, (size_t)e.tH(s0) * e.tW(s0), s0.Cs,
                                       op.d_w, op.relu, st));
                break;
            }
            case OP_LOGITS:
                PSEG_TRY(mfma_launch_logits(e, op, d_logits, d_probs, d_labels, d_labels_u8, st));
                break;
        }
        PSEG_HIP(hipGetLastError());
        PSEG_TRY(time_end(e, op, st, ev0));
    }
    if (d_margin && !e.margin_done) {
        if (!d_logits) return fail(PSEG_EINVAL, "margin map requested but neither the tail kernel nor a logits buffer provides it");
        launch_margin_from_logits(d_logits, (size_t)e.H * e.W, e.n_classes, d_margin, st);
    }
    e.cur_margin = nullptr;
    return PSEG_OK;
}

// `n` pages of one shape (contiguous in d_imgs, label maps contiguous in d_labels / d_labels_u8) through the bf16 graph with every
// activation tensor holding n page slots (lib/predictor.py:27-30 is a loop over pages; configs[2] = 32 per rank).  Layers whose
// kernel walks tiles of several slots take the whole batch in one launch -- at 1/4 and 1/8 resolution a page has 768 / 192 tiles
// for 256 CUs: a launch of one page ends in a partly filled round, starts with every workgroup fetching its first tile at once
// and, at 1/8 resolution, leaves a quarter of the chip idle -- the others run page by page on the slot's pointers.  Runs of
// per-page layers go page-major (a page's tensors stay in the caches between its layers), batched layers layer-major.
static int run_bf16_pages(Engine& e, const uint8_t* d_imgs, int n, int64_t* d_labels, uint8_t* d_labels_u8, hipStream_t st) {
    const size_t npx = (size_t)e.H * e.W;
    e.cur_margin = nullptr;
    e.margin_done = false;
    e.cur_logits = nullptr;
    e.cur_probs = nullptr;
    auto slot = [&](int p) {
        for (auto& t : e.tensors) t.d = t.base ? (char*)t.base + (size_t)p * t.page_bytes : nullptr;
        e.page = p;
        e.cur_labels = d_labels ? d_labels + (size_t)p * npx : nullptr;
        e.cur_labels_u8 = d_labels_u8 ? d_labels_u8 + (size_t)p * npx : nullptr;
    };
    auto launch = [&](Op& op) -> int {
        hipEvent_t ev0;
        PSEG_TRY(time_begin(e, op, st, &ev0));
        switch (op.type) {
            case OP_CONV: PSEG_TRY(mfma_launch_conv(e, op, st)); break;
            case OP_DECONV2: PSEG_TRY(mfma_launch_deconv2(e, op, st)); break;
            case OP_POOL: PSEG_TRY(mfma_launch_pool(e, op, st)); break;
            case OP_BN: {
                const Tensor& s0 = e.tensors[op.src0];
                PSEG_TRY(bn_infer_bf16((const uint16_t*)s0.d, (uint16_t*)e.tensors[op.dst].d, (size_t)e.tH(s0) * e.tW(s0), s0.Cs, op.d_w, op.relu, st));
                break;
            }
            case OP_LOGITS: PSEG_TRY(mfma_launch_logits(e, op, nullptr, nullptr, e.cur_labels, e.cur_labels_u8, st)); break;   // (the slot's label maps)
            default: return fail(PSEG_EUNSUPPORTED, "page units: unknown op type (layer %s)", op.layer.c_str());
        }
        PSEG_HIP(hipGetLastError());
        return time_end(e, op, st, ev0);
    };
    int rc = PSEG_OK;
    std::vector<Op*> live;
    for (auto& op : e.ops)
        if (!op.fused_away) live.push_back(&op);
    // graphs whose first layer is not fused into its consumer (unet, res_unet, 3-channel input) read the pre-processed page from the
    // input TENSOR: every slot's is written once, up front; the fused first layer reads the uint8 page itself (cur_img, set per slot below)
    bool input_tensor_read = false;
    for (auto& op : e.ops)
        if (!op.fused_away && (op.src0 == e.input_tensor || op.src1 == e.input_tensor) && op.fuse1 < 0) input_tensor_read = true;
    // (mfma_preprocess launches nothing where every reader of the input is a special first-layer kernel that takes the page bytes)
    if (input_tensor_read)
        for (int p = 0; p < n && rc == PSEG_OK; ++p) { slot(p); rc = mfma_preprocess(e, d_imgs + (size_t)p * npx * e.in_ch, st); }
    for (size_t i = 0; i < live.size() && rc == PSEG_OK;) {
        if (mfma_op_batchable(e, *live[i])) {
            slot(0);
            e.batch_pages = n;
            rc = launch(*live[i]);
            e.batch_pages = 0;
            ++i;
            continue;
        }
        size_t j = i;
        while (j < live.size() && !mfma_op_batchable(e, *live[j])) ++j;
        for (int p = 0; p < n && rc == PSEG_OK; ++p) {
            slot(p);
            e.cur_img = d_imgs + (size_t)p * npx * e.in_ch;                     // (the page a fused / special first layer reads)
            for (size_t k = i; k < j && rc == PSEG_OK; ++k) rc = launch(*live[k]);
        }
        i = j;
    }
    slot(0);
    e.cur_labels = nullptr;
    e.cur_labels_u8 = nullptr;
    return rc;
}

// pseg_predict_batch's device leg for a group of same-shape pages
int predict_device_pages(Engine& e, const uint8_t* d_imgs, int n, int H, int W, int64_t* d_labels, uint8_t* d_labels_u8, hipStream_t st) {
    PSEG_HIP(hipSetDevice(e.device));
    for (auto& p : e.params)
        if (!p.set) return fail(PSEG_EINVAL, "weight '%s' was never set", p.name.c_str());
    if (e.weights_dirty) PSEG_TRY(upload_weights(e));
    PSEG_TRY(set_canvas(e, H, W, st, n));
    return run_bf16_pages(e, d_imgs, n, d_labels, d_labels_u8, st);
}

// true when this engine's graph can run page units (run_bf16_pages): bf16 mode and at least one layer whose kernel takes several page
// slots in one launch (the fcn family's 1/4- and 1/8-resolution layers; the plain convs of unet / res_unet from 1/4 resolution down)
bool pages_capable(Engine& e) {
    if (e.mode != PSEG_MODE_BF16 || PSEG_KNOB("PSEG_NO_PAGE_BATCH")) return false;
    if (e.weights_dirty) return false;        // (plans exist after the first upload: the caller's first page goes alone)
    bool any = false;
    for (auto& op : e.ops)
        if (!op.fused_away) any |= mfma_op_batchable(e, op);
    return any;
}

int predict_device(Engine& e, const uint8_t* d_img, int H, int W, float* d_logits,
                   float* d_probs, int64_t* d_labels, uint8_t* d_labels_u8,
                   hipStream_t st, float* d_margin) {
    PSEG_HIP(hipSetDevice(e.device));
    for (auto& p : e.params)
        if (!p.set) return fail(PSEG_EINVAL, "weight '%s' was never set", p.name.c_str());
    if (e.weights_dirty) PSEG_TRY(upload_weights(e));
    PSEG_TRY(set_canvas(e, H, W, st));
    if (e.mode == PSEG_MODE_BF16)
        return run_bf16(e, d_img, d_logits, d_probs, d_labels, d_labels_u8, d_margin, st);
    if (d_margin && !d_logits) {
        PSEG_TRY(ensure((void**)&e.d_logits_tmp, &e.logits_tmp_bytes, (size_t)H * W * e.n_classes * sizeof(float)));
        d_logits = e.d_logits_tmp;
    }
    PSEG_TRY(run_exact(e, d_img, d_logits, d_probs, d_labels, d_labels_u8, st));
    if (d_margin) launch_margin_from_logits(d_logits, (size_t)H * W, e.n_classes, d_margin, st);
    return PSEG_OK;
}

// ---- page batches: copies of neighbouring pages overlap the compute of the current one -----------
struct BatchState {
    hipStream_t s_in = nullptr, s_out = nullptr;
    hipEvent_t up[2] = {nullptr, nullptr}, done[2] = {nullptr, nullptr}, down[2] = {nullptr, nullptr};
    uint8_t* d_img[2] = {nullptr, nullptr};
    void* d_lab[2] = {nullptr, nullptr};
    size_t img_bytes[2] = {0, 0}, lab_bytes[2] = {0, 0};
    // pinned staging ring for callers whose pages / label maps live in pageable memory: the page is copied into the
    // slot by the calling thread and leaves it by DMA; label maps arrive in the slot by DMA and are copied out by the
    // calling thread while the next page computes.  Buffers from pseg_host_alloc / pseg_host_register skip the ring.
    uint8_t* h_in[2] = {nullptr, nullptr};
    uint8_t* h_out[2] = {nullptr, nullptr};
    size_t h_in_bytes[2] = {0, 0}, h_out_bytes[2] = {0, 0};
};

static int ensure_pinned(uint8_t** p, size_t* cap, size_t bytes) {
    if (*cap >= bytes && *p) return PSEG_OK;
    if (*p) (void)hipHostFree(*p);
    *p = nullptr;
    *cap = 0;
    PSEG_HIP(hipHostMalloc((void**)p, bytes, hipHostMallocDefault));
    *cap = bytes;
    return PSEG_OK;
}

// true when `p` is page-locked host memory the runtime knows (hipHostMalloc / hipHostRegister): DMA goes straight to it
static bool is_pinned(const void* p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeHost;
}

static void batch_free(Engine& e) {
    auto* b = (BatchState*)e.batch;
    if (!b) return;
    for (int i = 0; i < 2; ++i) {
        if (b->up[i]) (void)hipEventDestroy(b->up[i]);
        if (b->done[i]) (void)hipEventDestroy(b->done[i]);
        if (b->down[i]) (void)hipEventDestroy(b->down[i]);
        free_dev((void*&)b->d_img[i]);
        free_dev(b->d_lab[i]);
        if (b->h_in[i]) (void)hipHostFree(b->h_in[i]);
        if (b->h_out[i]) (void)hipHostFree(b->h_out[i]);
    }
    if (b->s_in) (void)hipStreamDestroy(b->s_in);
    if (b->s_out) (void)hipStreamDestroy(b->s_out);
    delete b;
    e.batch = nullptr;
}

int predict_device_pages(Engine& e, const uint8_t* d_imgs, int n, int H, int W, int64_t* d_labels, uint8_t* d_labels_u8, hipStream_t st);
bool pages_capable(Engine& e);

// units of a page list (see predict_batch): runs of same-shape pages, at most `cap` per unit, sizes 1, 2, 4 ... at the head and
// ... 4, 2, 1 at the tail of the list
static void plan_units(int n, const int* H, const int* W, int cap, std::vector<int>& ub, std::vector<int>& ug) {
    for (int i = 0; i < n;) {
        int run = 1;
        while (i + run < n && H[i + run] == H[i] && W[i + run] == W[i]) ++run;
        std::vector<int> front, back;
        int left = run;
        if (cap > 1) {
            const bool ramp_up = i == 0, ramp_down = i + run == n;
            for (int g = 1; g < cap && left > 0 && (ramp_up || ramp_down); g *= 2) {
                if (ramp_up && left >= g) { front.push_back(g); left -= g; }
                if (ramp_down && left >= g) { back.push_back(g); left -= g; }
            }
        }
        std::vector<int> sizes = front;
        while (left > 0) { const int g = std::min(cap, left); sizes.push_back(g); left -= g; }
        for (auto it = back.rbegin(); it != back.rend(); ++it) sizes.push_back(*it);
        for (int g : sizes) { ub.push_back(i); ug.push_back(g); i += g; }
    }
}

static int predict_batch(Engine& e, int n, const uint8_t* const* imgs, const int* H, const int* W,
                         int64_t* const* labels, uint8_t* const* labels_u8) {
    PSEG_HIP(hipSetDevice(e.device));
    if (!e.batch) {
        auto* nb = new BatchState();
        e.batch = nb;
        PSEG_HIP(hipStreamCreateWithFlags(&nb->s_in, hipStreamNonBlocking));
        PSEG_HIP(hipStreamCreateWithFlags(&nb->s_out, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) {
            PSEG_HIP(hipEventCreateWithFlags(&nb->up[i], hipEventDisableTiming));
            PSEG_HIP(hipEventCreateWithFlags(&nb->done[i], hipEventDisableTiming));
            PSEG_HIP(hipEventCreateWithFlags(&nb->down[i], hipEventDisableTiming));
        }
    }
    auto* b = (BatchState*)e.batch;
    for (auto& p : e.params)
        if (!p.set) return fail(PSEG_EINVAL, "weight '%s' was never set", p.name.c_str());
    if (e.weights_dirty) PSEG_TRY(upload_weights(e));      // (the plans decide whether pages travel as units)
    for (int i = 0; i < n; ++i) {
        if (H[i] <= 0 || W[i] <= 0 || !imgs[i] || (labels && !labels[i]) || (labels_u8 && !labels_u8[i]))
            return fail(PSEG_EINVAL, "page %d: empty shape or NULL buffer", i);
    }
    std::vector<char> in_pinned(n), out_pinned(n);
    for (int i = 0; i < n; ++i) {
        in_pinned[i] = is_pinned(imgs[i]);
        out_pinned[i] = (!labels || is_pinned(labels[i])) && (!labels_u8 || is_pinned(labels_u8[i]));
    }
    // Units: runs of consecutive pages of one shape go through the graph together (run_bf16_pages: every tensor holds a page slot
    // per page of the unit, the low-resolution layers take all slots in one launch); everything else is a unit of one page.
    // PSEG_BATCH_PAGES caps a unit (default 8: 7 GB of activations at 2048x1536; 1 = page by page as before).
    int cap = 8;
    if (const char* ev = PSEG_KNOB("PSEG_BATCH_PAGES")) cap = std::max(1, std::min(64, atoi(ev)));
    // ... and the copies of a unit overlap the compute of its NEIGHBOURS only: a list shorter than four units would wait for its first
    // upload and its last download with nothing beside them (8 pages as one unit: 0.70 ms per page against 0.49 page by page)
    cap = std::min(cap, std::max(1, n / 4));
    if (!pages_capable(e)) cap = 1;
    if (cap > 1) {                          // ... and by what the device has room for (the largest page of the list decides)
        int hm = 0, wm = 0;
        for (int i = 0; i < n; ++i)
            if ((size_t)H[i] * W[i] > (size_t)hm * wm) { hm = H[i]; wm = W[i]; }
        cap = fit_page_slots(e, hm, wm, cap);
    }
    // Unit sizes.  The upload of the FIRST unit and the download of the LAST one have no compute beside them: a list cut into
    // equal units of 8 pages waits 8 page uploads at its head and 8 downloads at its tail (32 pages: 2 of 14 ms).  So a run of same-shape
    // pages that opens the list starts with units of 1, 2, 4, ... pages, one that closes it ends ... 4, 2, 1 (every unit's copies
    // still fit under its neighbour's compute: a page computes for 0.4 ms and travels for 0.12), the middle goes in units of `cap`.
    // Same box, 32 / 8 pages of 2048x1536, ms per page, ramped against equal units: pinned uint8 0.410-0.445 vs 0.430 / 0.458 vs 0.466,
    // pageable arrays through the ring 0.456 vs 0.57 / 0.51-0.52 vs 0.55 (tools/gpu_r05_hostpath.sh).
    std::vector<int> ub, ug;                 // first page, page count of every unit
    plan_units(n, H, W, cap, ub, ug);
    const int nu = (int)ub.size();
    auto upx = [&](int u) { return (size_t)H[ub[u]] * W[ub[u]]; };
    auto lab_off8 = [&](int u, int k) { return (size_t)k * upx(u) * 8; };                                   // int64 map of page k of the unit
    auto lab_off1 = [&](int u, int k) { return (labels ? (size_t)ug[u] * upx(u) * 8 : 0) + (size_t)k * upx(u); };   // its uint8 map
    auto unit_out_bytes = [&](int u) { return (size_t)ug[u] * upx(u) * ((labels ? 8 : 0) + (labels_u8 ? 1 : 0)); };
    auto unit_pinned = [&](int u, const std::vector<char>& v) { bool all = true; for (int k = 0; k < ug[u]; ++k) all = all && v[ub[u] + k]; return all; };
    auto upload = [&](int u) -> int {        // unit u -> slot u % 2 (its previous compute has been waited for)
        const int s = u & 1;
        const size_t npx = upx(u), pb = npx * e.in_ch;
        PSEG_TRY(ensure((void**)&b->d_img[s], &b->img_bytes[s], pb * ug[u]));
        PSEG_TRY(ensure(&b->d_lab[s], &b->lab_bytes[s], unit_out_bytes(u) + 16));
        PSEG_HIP(hipStreamWaitEvent(b->s_in, b->done[s], 0));      // slot input consumed (no-op before first record)
        const bool pinned = unit_pinned(u, in_pinned);
        if (!pinned) {
            // the ring slot was last read by the upload of unit u-2, recorded in up[s]
            PSEG_HIP(hipEventSynchronize(b->up[s]));
            PSEG_TRY(ensure_pinned(&b->h_in[s], &b->h_in_bytes[s], pb * ug[u]));
            for (int k = 0; k < ug[u]; ++k) memcpy(b->h_in[s] + (size_t)k * pb, imgs[ub[u] + k], pb);
            PSEG_HIP(hipMemcpyAsync(b->d_img[s], b->h_in[s], pb * ug[u], hipMemcpyHostToDevice, b->s_in));
        } else {
            for (int k = 0; k < ug[u]; ++k)
                PSEG_HIP(hipMemcpyAsync(b->d_img[s] + (size_t)k * pb, imgs[ub[u] + k], pb, hipMemcpyHostToDevice, b->s_in));
        }
        PSEG_HIP(hipEventRecord(b->up[s], b->s_in));
        return PSEG_OK;
    };
    auto compute = [&](int u) -> int {
        const int s = u & 1, i0 = ub[u];
        // a canvas change re-allocates / clears the activation tensors: the previous unit must have left them
        if (round_up(H[i0], 32) != e.Hp || round_up(W[i0], 32) != e.Wp || ug[u] > e.pages) PSEG_HIP(hipStreamSynchronize(e.stream));
        PSEG_HIP(hipStreamWaitEvent(e.stream, b->up[s], 0));
        PSEG_HIP(hipStreamWaitEvent(e.stream, b->down[s], 0));      // slot output of unit u-2 has left
        int64_t* dl = labels ? (int64_t*)b->d_lab[s] : nullptr;
        uint8_t* du = labels_u8 ? (uint8_t*)b->d_lab[s] + lab_off1(u, 0) : nullptr;
        if (ug[u] > 1) PSEG_TRY(predict_device_pages(e, b->d_img[s], ug[u], H[i0], W[i0], dl, du, e.stream));
        else PSEG_TRY(predict_device(e, b->d_img[s], H[i0], W[i0], nullptr, nullptr, dl, du, e.stream, nullptr));
        PSEG_HIP(hipEventRecord(b->done[s], e.stream));
        return PSEG_OK;
    };
    auto download = [&](int u) -> int {
        const int s = u & 1;
        const size_t npx = upx(u);
        PSEG_HIP(hipStreamWaitEvent(b->s_out, b->done[s], 0));
        if (unit_pinned(u, out_pinned)) {
            for (int k = 0; k < ug[u]; ++k) {
                if (labels) PSEG_HIP(hipMemcpyAsync(labels[ub[u] + k], (uint8_t*)b->d_lab[s] + lab_off8(u, k), npx * 8, hipMemcpyDeviceToHost, b->s_out));
                if (labels_u8) PSEG_HIP(hipMemcpyAsync(labels_u8[ub[u] + k], (uint8_t*)b->d_lab[s] + lab_off1(u, k), npx, hipMemcpyDeviceToHost, b->s_out));
            }
        } else {   // all maps of the unit in one DMA into the ring slot (its previous content was copied out by finish(u - 2))
            PSEG_TRY(ensure_pinned(&b->h_out[s], &b->h_out_bytes[s], unit_out_bytes(u)));
            PSEG_HIP(hipMemcpyAsync(b->h_out[s], b->d_lab[s], unit_out_bytes(u), hipMemcpyDeviceToHost, b->s_out));
        }
        PSEG_HIP(hipEventRecord(b->down[s], b->s_out));
        return PSEG_OK;
    };
    auto finish = [&](int u) -> int {        // pageable destination: ring slot -> caller's arrays, on the calling thread
        if (unit_pinned(u, out_pinned)) return PSEG_OK;
        const int s = u & 1;
        const size_t npx = upx(u);
        PSEG_HIP(hipEventSynchronize(b->down[s]));
        for (int k = 0; k < ug[u]; ++k) {
            if (labels) memcpy(labels[ub[u] + k], b->h_out[s] + lab_off8(u, k), npx * 8);
            if (labels_u8) memcpy(labels_u8[ub[u] + k], b->h_out[s] + lab_off1(u, k), npx);
        }
        return PSEG_OK;
    };
    // every way out -- also an error return in the middle -- ends with the three streams drained: copies from / to the caller's
    // host arrays and the ring slots must not be in flight when the caller gets its buffers back
    struct Drain { hipStream_t a, b, c; ~Drain() { (void)hipStreamSynchronize(a); (void)hipStreamSynchronize(b); (void)hipStreamSynchronize(c); } } drain{b->s_in, e.stream, b->s_out};
    // a reallocation of a slot must not race with work still using it: size every slot for the largest unit up front
    size_t max_in = 0, max_out = 0;
    for (int u = 0; u < nu; ++u) { max_in = std::max(max_in, upx(u) * e.in_ch * ug[u]); max_out = std::max(max_out, unit_out_bytes(u)); }
    for (int s = 0; s < 2; ++s) {
        PSEG_TRY(ensure((void**)&b->d_img[s], &b->img_bytes[s], max_in));
        PSEG_TRY(ensure(&b->d_lab[s], &b->lab_bytes[s], max_out + 16));
    }
    bool any_in = false, any_out = false;
    for (int i = 0; i < n; ++i) { any_in |= !in_pinned[i]; any_out |= !out_pinned[i]; }
    for (int s = 0; s < 2 && (any_in || any_out); ++s) {
        PSEG_HIP(hipEventSynchronize(b->up[s]));
        PSEG_HIP(hipEventSynchronize(b->down[s]));
        if (any_in) PSEG_TRY(ensure_pinned(&b->h_in[s], &b->h_in_bytes[s], max_in));
        if (any_out) PSEG_TRY(ensure_pinned(&b->h_out[s], &b->h_out_bytes[s], max_out));
    }
    if (nu > 0) { PSEG_TRY(upload(0)); PSEG_TRY(compute(0)); }
    for (int u = 0; u < nu; ++u) {
        if (u + 1 < nu) { PSEG_TRY(upload(u + 1)); PSEG_TRY(compute(u + 1)); }
        PSEG_TRY(download(u));
        if (u > 0) PSEG_TRY(finish(u - 1));
    }
    if (nu > 0) PSEG_TRY(finish(nu - 1));
    PSEG_HIP(hipStreamSynchronize(b->s_out));
    return engine_status(e, e.stream);     // (waits for the stream; a give-up of conv_sp_kernel in any unit is an error here)
}

}  // namespace pseg

// =============================================================================================
// C ABI
// =============================================================================================
using namespace pseg;


extern "C" {

int pseg_abi_version(void) { return PSEG_ABI_VERSION; }

const char* pseg_last_error(void) { return last_error().c_str(); }

int pseg_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int pseg_create(int arch, int n_classes, int in_channels, int device, int mode,
                pseg_engine** out) {
    return pseg_create_ex(arch, n_classes, in_channels, device, mode, 0u, out);
}

int pseg_create_ex(int arch, int n_classes, int in_channels, int device, int mode, unsigned flags,
                   pseg_engine** out) {
    return pseg::create_engine(arch, n_classes, in_channels, device, mode, flags, nullptr, out);
}

int pseg_create_plan(int arch, int n_classes, int in_channels, int device, int mode, unsigned flags, const char* switches,
                     pseg_engine** out) {
    return pseg::create_engine(arch, n_classes, in_channels, device, mode, flags, nullptr, out, switches);
}

const char* pseg_env_knobs(void) {
    static const std::string joined = [] { std::string j; for (const char* n : k_env_knobs) { if (!j.empty()) j += "\n"; j += n; } return j; }();
    return joined.c_str();
}

}  // extern "C"

// `inherit`: the snapshot of a parent engine (the float32 companion of the label-exact mode is created in the middle of
// its parent's predict call and must not see another environment than its parent); nullptr = snapshot the environment now
int pseg::create_engine(int arch, int n_classes, int in_channels, int device, int mode, unsigned flags,
                        std::shared_ptr<const KnobSnap> inherit, pseg_engine** out, const char* plan) {
    if (!out) return fail(PSEG_EINVAL, "out is NULL");
    if (flags & ~(unsigned)PSEG_FLAG_BATCHNORM) return fail(PSEG_EINVAL, "unknown flag bits 0x%x", flags);
    *out = nullptr;
    if (n_classes < 1 || n_classes > 256) return fail(PSEG_EINVAL, "n_classes %d out of range", n_classes);
    if (in_channels != 1 && in_channels != 3) return fail(PSEG_EINVAL, "in_channels must be 1 or 3");
    if (mode != PSEG_MODE_F32_EXACT && mode != PSEG_MODE_BF16) return fail(PSEG_EINVAL, "bad mode %d", mode);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(PSEG_EHIP, "no HIP device visible: libpseg has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(PSEG_EINVAL, "device %d of %d", device, ndev);
    PSEG_HIP(hipSetDevice(device));
    auto* h = new pseg_engine();
    Engine& e = h->e;
    e.knobs = inherit ? inherit : knobs_snapshot(plan);   // the knobs are read here, once per engine, never per launch
    KnobScope ks(e);
    e.arch = arch;
    e.n_classes = n_classes;
    e.in_ch = in_channels;
    e.device = device;
    e.mode = mode;
    e.flags = flags;
    int rc = build_graph(e);
    if (rc == PSEG_OK && mode == PSEG_MODE_BF16) rc = mfma_plan_graph(e);
    if (rc != PSEG_OK) { delete h; return rc; }
    if (hipStreamCreateWithFlags(&e.stream, hipStreamNonBlocking) != hipSuccess) {
        delete h;
        return fail(PSEG_EHIP, "hipStreamCreate failed");
    }
    float lut[256];
    for (int i = 0; i < 256; ++i) lut[i] = (float)i / 255.0f;  // lib/architecture.py:67-68
    if (hipMalloc((void**)&e.d_lut, sizeof lut) != hipSuccess ||
        hipMemcpy(e.d_lut, lut, sizeof lut, hipMemcpyHostToDevice) != hipSuccess) {
        delete h;
        return fail(PSEG_EHIP, "LUT upload failed");
    }
    (void)hipDeviceSynchronize();
    *out = h;
    return PSEG_OK;
}

extern "C" {

int pseg_destroy(pseg_engine* h) {
    if (!h) return PSEG_OK;
    Engine& e = h->e;
    (void)hipSetDevice(e.device);
    if (e.stream) (void)hipStreamSynchronize(e.stream);
    train_free(e);
    dist_free(e);
    exact_free(e);
    chain_free(e);
    batch_free(e);
    for (auto& t : e.tensors) { free_dev(t.base); t.d = nullptr; }
    for (auto& op : e.ops) {
        free_dev((void*&)op.d_w);
        free_dev((void*&)op.d_b);
        free_dev((void*&)op.d_wrem);
        mfma_free_op(op);
    }
    free_dev((void*&)e.d_lut);
    free_dev((void*&)e.d_sp_err);
    if (e.h_sp_err) { (void)hipHostFree(e.h_sp_err); e.h_sp_err = nullptr; }
    free_dev((void*&)e.d_logits_tmp);
    free_dev((void*&)e.d_img_stage);
    free_dev((void*&)e.d_lab_stage);
    free_dev((void*&)e.d_prob_stage);
    free_dev((void*&)e.d_logit_stage);
    for (auto& s : e.slots)
        for (auto& pr : s.pending) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (auto ev : e.event_pool) (void)hipEventDestroy(ev);
    if (e.stream) (void)hipStreamDestroy(e.stream);
    delete h;
    return PSEG_OK;
}

int pseg_num_weights(const pseg_engine* h) { return h ? (int)h->e.params.size() : 0; }

int pseg_weight_info(const pseg_engine* h, int index, char* name, size_t name_cap,
                     int64_t shape[4], int* ndim) {
    if (!h || index < 0 || index >= (int)h->e.params.size()) return fail(PSEG_EINVAL, "bad weight index %d", index);
    const Param& p = h->e.params[index];
    if (name && name_cap) {
        strncpy(name, p.name.c_str(), name_cap - 1);
        name[name_cap - 1] = 0;
    }
    if (shape) for (int i = 0; i < 4; ++i) shape[i] = i < p.ndim ? p.shape[i] : 1;
    if (ndim) *ndim = p.ndim;
    return PSEG_OK;
}

static Param* find_param(Engine& e, const char* name) {
    for (auto& p : e.params)
        if (p.name == name) return &p;
    return nullptr;
}

int pseg_set_weights(pseg_engine* h, const char* name, const float* data, const int64_t* shape,
                     int ndim) {
    if (!h || !name || !data || !shape) return fail(PSEG_EINVAL, "NULL argument");
    KnobScope knob_scope(h->e);
    Param* p = find_param(h->e, name);
    if (!p) return fail(PSEG_ENOTFOUND, "no weight named '%s'", name);
    if (ndim != p->ndim) return fail(PSEG_EINVAL, "weight '%s': rank %d, expected %d", name, ndim, p->ndim);
    for (int i = 0; i < ndim; ++i)
        if (shape[i] != p->shape[i])
            return fail(PSEG_EINVAL, "weight '%s': dim %d is %lld, expected %lld", name, i,
                        (long long)shape[i], (long long)p->shape[i]);
    std::copy(data, data + p->host.size(), p->host.begin());
    p->set = true;
    h->e.weights_dirty = true;
    h->e.exact_dirty = true;
    return PSEG_OK;
}

int pseg_get_weights(const pseg_engine* h, const char* name, float* out, int64_t count) {
    if (!h || !name || !out) return fail(PSEG_EINVAL, "NULL argument");
    Param* p = find_param(const_cast<Engine&>(h->e), name);
    if (!p) return fail(PSEG_ENOTFOUND, "no weight named '%s'", name);
    if (count != (int64_t)p->host.size()) return fail(PSEG_EINVAL, "weight '%s' has %zu elements", name, p->host.size());
    if (h->e.train) PSEG_TRY(train_sync_weights_to_host(const_cast<Engine&>(h->e)));
    std::copy(p->host.begin(), p->host.end(), out);
    return PSEG_OK;
}

int pseg_predict_device(pseg_engine* h, const uint8_t* d_img, int H, int W, float* d_logits,
                        float* d_probs, int64_t* d_labels, uint8_t* d_labels_u8, void* stream) {
    if (!h || !d_img) return fail(PSEG_EINVAL, "NULL argument");
    KnobScope knob_scope(h->e);
    hipStream_t st = stream ? (hipStream_t)stream : h->e.stream;
    return predict_device(h->e, d_img, H, W, d_logits, d_probs, d_labels, d_labels_u8, st, nullptr);
}

int pseg_predict_pages_device(pseg_engine* h, const uint8_t* d_imgs, int n, int H, int W, int64_t* d_labels, uint8_t* d_labels_u8, void* stream) {
    if (!h || !d_imgs || n < 1 || (!d_labels && !d_labels_u8)) return fail(PSEG_EINVAL, "NULL argument / no pages");
    KnobScope knob_scope(h->e);
    Engine& e = h->e;
    hipStream_t st = stream ? (hipStream_t)stream : e.stream;
    if (H <= 0 || W <= 0) return fail(PSEG_EINVAL, "empty page %dx%d", H, W);
    PSEG_HIP(hipSetDevice(e.device));
    for (auto& p : e.params)
        if (!p.set) return fail(PSEG_EINVAL, "weight '%s' was never set", p.name.c_str());
    if (e.weights_dirty) PSEG_TRY(upload_weights(e));
    const size_t npx = (size_t)H * W;
    if (n > 1 && pages_capable(e)) {
        int cap = 16;                         // page slots per unit of a device-resident batch (14 GB of activations at 2048x1536; 8 and 16 measure alike)
        if (const char* ev = PSEG_KNOB("PSEG_BATCH_PAGES")) cap = std::max(1, std::min(64, atoi(ev)));
        cap = fit_page_slots(e, H, W, std::min(cap, n));
        for (int i = 0; i < n;) {
            const int g = std::min(cap, n - i);
            const int rc = g > 1 ? predict_device_pages(e, d_imgs + (size_t)i * npx * e.in_ch, g, H, W, d_labels ? d_labels + (size_t)i * npx : nullptr,
                                                        d_labels_u8 ? d_labels_u8 + (size_t)i * npx : nullptr, st)
                                 : predict_device(e, d_imgs + (size_t)i * npx * e.in_ch, H, W, nullptr, nullptr, d_labels ? d_labels + (size_t)i * npx : nullptr,
                                                  d_labels_u8 ? d_labels_u8 + (size_t)i * npx : nullptr, st, nullptr);
            if (rc == PSEG_ENOMEM && cap > 1) { cap = (cap + 1) / 2; continue; }     // (set_canvas left the engine without a canvas: half the slots)
            PSEG_TRY(rc);
            i += g;
        }
        return PSEG_OK;
    }
    for (int i = 0; i < n; ++i)
        PSEG_TRY(predict_device(e, d_imgs + (size_t)i * npx * e.in_ch, H, W, nullptr, nullptr, d_labels ? d_labels + (size_t)i * npx : nullptr,
                                d_labels_u8 ? d_labels_u8 + (size_t)i * npx : nullptr, st, nullptr));
    return PSEG_OK;
}

int pseg_predict(pseg_engine* h, const uint8_t* img, int H, int W, float* logits, float* probs,
                 int64_t* labels) {
    if (!h || !img) return fail(PSEG_EINVAL, "NULL argument");
    KnobScope knob_scope(h->e);
    if (H <= 0 || W <= 0) return fail(PSEG_EINVAL, "empty page %dx%d", H, W);
    Engine& e = h->e;
    PSEG_HIP(hipSetDevice(e.device));
    const size_t npx = (size_t)H * W, C = e.n_classes;
    PSEG_TRY(ensure((void**)&e.d_img_stage, &e.img_stage_bytes, npx * e.in_ch));
    if (labels) PSEG_TRY(ensure((void**)&e.d_lab_stage, &e.lab_stage_bytes, npx * 8));
    if (probs) PSEG_TRY(ensure((void**)&e.d_prob_stage, &e.prob_stage_bytes, npx * C * 4));
    if (logits) PSEG_TRY(ensure((void**)&e.d_logit_stage, &e.logit_stage_bytes, npx * C * 4));
    PSEG_HIP(hipMemcpyAsync(e.d_img_stage, img, npx * e.in_ch, hipMemcpyHostToDevice, e.stream));
    PSEG_TRY(predict_device(e, e.d_img_stage, H, W, logits ? e.d_logit_stage : nullptr,
                            probs ? e.d_prob_stage : nullptr, labels ? e.d_lab_stage : nullptr,
                            nullptr, e.stream, nullptr));
    if (logits) PSEG_HIP(hipMemcpyAsync(logits, e.d_logit_stage, npx * C * 4, hipMemcpyDeviceToHost, e.stream));
    if (probs) PSEG_HIP(hipMemcpyAsync(probs, e.d_prob_stage, npx * C * 4, hipMemcpyDeviceToHost, e.stream));
    if (labels) PSEG_HIP(hipMemcpyAsync(labels, e.d_lab_stage, npx * 8, hipMemcpyDeviceToHost, e.stream));
    return engine_status(e, e.stream);
}

int pseg_engine_status(pseg_engine* h, void* stream) {
    if (!h) return fail(PSEG_EINVAL, "NULL engine");
    KnobScope knob_scope(h->e);
    PSEG_HIP(hipSetDevice(h->e.device));
    return engine_status(h->e, stream ? (hipStream_t)stream : h->e.stream);
}

int pseg_engine_trim(pseg_engine* h) {
    if (!h) return fail(PSEG_EINVAL, "NULL engine");
    Engine& e = h->e;
    PSEG_HIP(hipSetDevice(e.device));
    PSEG_HIP(hipDeviceSynchronize());
    for (auto& t : e.tensors) { free_dev(t.base); t.d = nullptr; t.bytes = 0; t.page_bytes = 0; }
    for (auto& op : e.ops) mfma_trim_op(op);                 // (the skip-logits planes grow with canvas x slots too)
    e.Hp = e.Wp = 0;
    e.pages = 1;
    free_dev((void*&)e.d_logits_tmp); e.logits_tmp_bytes = 0;
    return PSEG_OK;
}

int pseg_predict_batch(pseg_engine* h, int n_pages, const uint8_t* const* imgs, const int* H, const int* W,
                       int64_t* const* labels, uint8_t* const* labels_u8) {
    if (!h || n_pages < 0 || (n_pages > 0 && (!imgs || !H || !W))) return fail(PSEG_EINVAL, "bad argument");
    KnobScope knob_scope(h->e);
    if (!labels && !labels_u8) return fail(PSEG_EINVAL, "no output requested");
    return predict_batch(h->e, n_pages, imgs, H, W, labels, labels_u8);
}

int pseg_batch_units(int n_pages, const int* H, const int* W, int cap, int* unit_first, int* unit_count, int max_units) {
    if (n_pages < 0 || (n_pages > 0 && (!H || !W)) || cap < 1) return fail(PSEG_EINVAL, "bad argument");
    std::vector<int> ub, ug;
    plan_units(n_pages, H, W, cap, ub, ug);
    if ((int)ub.size() > max_units && (unit_first || unit_count)) return fail(PSEG_EINVAL, "%zu units, room for %d", ub.size(), max_units);
    for (size_t u = 0; u < ub.size(); ++u) {
        if (unit_first) unit_first[u] = ub[u];
        if (unit_count) unit_count[u] = ug[u];
    }
    return (int)ub.size();
}

int pseg_host_alloc(void** p, size_t bytes) {
    if (!p) return fail(PSEG_EINVAL, "NULL argument");
    *p = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(PSEG_EHIP, "no HIP device visible: libpseg has no CPU fallback");
    PSEG_HIP(hipHostMalloc(p, bytes ? bytes : 1, hipHostMallocDefault));
    return PSEG_OK;
}

int pseg_host_free(void* p) {
    if (p) PSEG_HIP(hipHostFree(p));
    return PSEG_OK;
}

int pseg_host_register(void* p, size_t bytes) {
    if (!p || !bytes) return fail(PSEG_EINVAL, "NULL / empty buffer");
    PSEG_HIP(hipHostRegister(p, bytes, hipHostRegisterDefault));
    return PSEG_OK;
}

int pseg_host_unregister(void* p) {
    if (p) PSEG_HIP(hipHostUnregister(p));
    return PSEG_OK;
}

// bf16 -> f32 helper for activation read-back
static inline float bf16_to_f32(uint16_t v) {
    uint32_t u = (uint32_t)v << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

int pseg_get_activation(pseg_engine* h, const char* layer, float* out, int64_t cap, int dims[3]) {
    if (!h || !layer) return fail(PSEG_EINVAL, "NULL argument");
    KnobScope knob_scope(h->e);
    Engine& e = h->e;
    PSEG_HIP(hipSetDevice(e.device));
    for (auto& t : e.tensors) {
        if (t.name != layer) continue;
        if (!t.d || e.Hp == 0) return fail(PSEG_EINVAL, "no predict call has run yet");
        if (t.fused) return fail(PSEG_EUNSUPPORTED, "layer '%s' is fused into its consumer and never materialised", layer);
        if (t.relu_stored) return fail(PSEG_EUNSUPPORTED, "layer '%s' is fused with its readers' pre-activation ReLU: stored as max(x, 0) (PSEG_NO_RELU_FWD=1 keeps the tensor)", layer);
        const int H = e.tH(t), W = e.tW(t);
        if (dims) { dims[0] = H; dims[1] = W; dims[2] = t.C; }
        const int64_t n = (int64_t)H * W * t.C;
        if (!out) return PSEG_OK;
        if (cap < n) return fail(PSEG_EINVAL, "activation '%s' needs %lld floats", layer, (long long)n);
        PSEG_HIP(hipStreamSynchronize(e.stream));
        if (e.mode == PSEG_MODE_F32_EXACT) {
            PSEG_HIP(hipMemcpy(out, t.d, (size_t)n * 4, hipMemcpyDeviceToHost));
        } else {
            std::vector<uint16_t> tmp((size_t)H * W * t.Cs);
            PSEG_HIP(hipMemcpy(tmp.data(), t.d, tmp.size() * 2, hipMemcpyDeviceToHost));
            for (int64_t p = 0; p < (int64_t)H * W; ++p)
                for (int c = 0; c < t.C; ++c) out[p * t.C + c] = bf16_to_f32(tmp[(size_t)p * t.Cs + c]);
        }
        return PSEG_OK;
    }
    return fail(PSEG_ENOTFOUND, "no layer named '%s'", layer);
}

void* pseg_engine_stream(pseg_engine* h) { return h ? (void*)h->e.stream : nullptr; }

double pseg_flops_per_pixel(const pseg_engine* h) {
    if (!h) return 0;
    double f = 0;
    for (auto& op : h->e.ops) f += op.flops_per_canvas_px;
    return f;
}

int pseg_timing_enable(pseg_engine* h, int on) {
    if (!h) return fail(PSEG_EINVAL, "NULL engine");
    h->e.timing = on != 0;
    return PSEG_OK;
}

int pseg_timing_reset(pseg_engine* h) {
    if (!h) return fail(PSEG_EINVAL, "NULL engine");
    PSEG_TRY(timing_collect(h->e));
    for (auto& s : h->e.slots) { s.total_ms = 0; s.launches = 0; }
    return PSEG_OK;
}

int pseg_timing_num_slots(const pseg_engine* h) { return h ? (int)h->e.slots.size() : 0; }

int pseg_timing_get(pseg_engine* h, int slot, char* name, size_t name_cap, double* total_ms,
                    int64_t* launches, double* flops) {
    if (!h || slot < 0 || slot >= (int)h->e.slots.size()) return fail(PSEG_EINVAL, "bad slot %d", slot);
    PSEG_HIP(hipSetDevice(h->e.device));
    PSEG_TRY(timing_collect(h->e));
    const TimingSlot& s = h->e.slots[slot];
    if (name && name_cap) {
        strncpy(name, s.name.c_str(), name_cap - 1);
        name[name_cap - 1] = 0;
    }
    if (total_ms) *total_ms = s.total_ms;
    if (launches) *launches = s.launches;
    if (flops) *flops = s.flops;
    return PSEG_OK;
}

}  // extern "C"
