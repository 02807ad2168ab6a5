// pseg_train.hip -- train step of the FCN (float32 engine): forward (pseg_engine.hip) -> sparse
// softmax cross-entropy + metrics -> backward -> per-tensor clip-by-norm -> Keras-formulation Adam.
//
// Reference semantics restated:
//   lib/metrics.py:8-9     loss      = mean over pixels of logsumexp(z) - z[y]
//   lib/metrics.py:12-17   accuracy  = mean(argmax(z) == y)
//   lib/metrics.py:60-85   jacard / dice with the +100 smoothing, per class over (H,W), mean over classes
//   lib/network.py:90-104  optimizer(lr, clipnorm): per-tensor clip_by_norm (TF2.5 `clipnorm`), then Adam
//   lib/architecture.py:83 Keras Adam: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v EMA; p -= lr_t*m/(sqrt(v)+eps), eps=1e-7
// Round-1 scope: fcn / fcn_skip graphs (stride-1 convs, k2s2 transposed convs, 2x2 max-pool,
// concat); correctness first -- these are plain float32 VALU kernels, not tuned.
// All parameter gradients live in ONE flat device buffer (plus the metric accumulators) so that
// data-parallel training needs a single RCCL all-reduce.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "pseg_common.h"
#include "pseg_wgrad.h"

namespace pseg {

constexpr int COT = 16;

struct TrainState {
    float beta1 = 0.9f, beta2 = 0.999f, eps = 1e-7f, clipnorm = 1.0f, clipvalue = 0.0f;
    int optimizer = PSEG_OPT_ADAM;   // lib/architecture.py:71-90
    int loss_kind = PSEG_LOSS_CE;    // lib/metrics.py:116-133
    double m_schedule = 1.0;         // Nadam's running product of the momentum schedule
    bool state_init = false;         // Adagrad: accumulators start at 0.1
    int64_t step = 0;
    // flat buffers: [params in e.params order][metrics: loss, correct, I_c (C), S_c (C)]
    float* d_grad = nullptr;
    float* d_m = nullptr;
    float* d_v = nullptr;
    float* d_norm = nullptr;     // per-parameter sum of squares
    float* d_part = nullptr;     // [slice][SUMSQ_MAXB] per-block partial sums of sumsq_kernel
    int* d_slice_nblk = nullptr; // per slice: blocks its sumsq_kernel ran / parameter index it belongs to
    int* d_slice_pi = nullptr;
    int nslices = 0;
    std::vector<int64_t> off;    // per param offset in the flat buffers
    int64_t nparam = 0, nflat = 0;
    std::vector<float*> tgrad;   // per tensor gradient (canvas dims), lazily sized
    std::vector<size_t> tbytes;
    float* d_wd = nullptr;       // scratch: transformed weights for dgrad
    size_t wd_bytes = 0;
    float* d_wdrem = nullptr;    // scratch: the left-over output channels of those weights as shifted copies (conv_xb_kernel REM), rebuilt per launch
    size_t wdrem_bytes = 0;
    float* d_wpart = nullptr;    // scratch: per-strip partial weight / bias gradients of the layer being reduced (deterministic sums)
    size_t wpart_bytes = 0;
    float* d_mpart = nullptr;    // scratch: per-block partial loss / metric sums
    size_t mpart_bytes = 0;
    void* d_bpart = nullptr;     // scratch: per-block partial bias gradients of a transposed conv (float64)
    size_t bpart_bytes = 0;
    float* d_tmp = nullptr;      // scratch: data gradient at the conv's input extent (upsampled / pre-activation sources)
    float* d_tmp2 = nullptr;     // scratch: zero-dilated output gradient (stride-2 convs)
    size_t tmp_bytes = 0, tmp2_bytes = 0;
    uint32_t drop_seed = 0x1234u;
    int64_t fwd_count = 0;       // training forwards so far: the Dropout mask changes every step
    float* d_logits = nullptr;
    float* d_dlogits = nullptr;
    uint8_t* d_mask = nullptr;
    uint8_t* d_img = nullptr;
    size_t logits_bytes = 0, mask_bytes = 0, img_bytes = 0;
    int H = 0, W = 0;
    // weight gradients run on a stream of their own beside the data gradients (both only read dY; see train_forward_backward)
    hipStream_t wstream = nullptr;
    hipEvent_t ev_dy = nullptr, ev_wdone = nullptr;
};

static TrainState* TS(Engine& e) { return (TrainState*)e.train; }

void train_free(Engine& e) {
    TrainState* t = TS(e);
    if (!t) return;
    (void)hipFree(t->d_grad); (void)hipFree(t->d_m); (void)hipFree(t->d_v); (void)hipFree(t->d_norm);
    (void)hipFree(t->d_part); (void)hipFree(t->d_slice_nblk); (void)hipFree(t->d_slice_pi);
    for (auto p : t->tgrad) (void)hipFree(p);
    (void)hipFree(t->d_wd); (void)hipFree(t->d_wdrem); (void)hipFree(t->d_logits); (void)hipFree(t->d_dlogits);
    (void)hipFree(t->d_wpart); (void)hipFree(t->d_mpart); (void)hipFree(t->d_bpart);
    (void)hipFree(t->d_mask); (void)hipFree(t->d_img); (void)hipFree(t->d_tmp); (void)hipFree(t->d_tmp2);
    if (t->wstream) { (void)hipStreamSynchronize(t->wstream); (void)hipStreamDestroy(t->wstream); }
    if (t->ev_dy) (void)hipEventDestroy(t->ev_dy);
    if (t->ev_wdone) (void)hipEventDestroy(t->ev_wdone);
    delete t;
    e.train = nullptr;
}

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
// acc layout: [0] sum loss, [1] count correct, [2..2+C) intersection_c, [2+C..2+2C) sum_c
// Deterministic: a thread's terms are reduced inside its wave by a butterfly, the four waves in order, and the block's
// sums go to ITS row of `part` ([block][2 + 2C]); metrics_final_kernel adds the rows in a fixed tree.  (LDS and global float
// atomics made the last bits of the reported loss -- and with dice / jaccard losses the gradient -- depend on arrival order.)
__global__ __launch_bounds__(256) void ce_metrics_kernel(const float* logits, const uint8_t* labels, int n, int C, float inv_n,
                                                         float* dlogits, float* part) {
    __shared__ float ws[4][2 + 2 * PSEG_MAXC];
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool live = p < n;
    const float* z = logits + (size_t)(live ? p : 0) * C;
    const int y = live ? labels[p] : 0;
    float m = z[0];
    int am = 0;
    for (int c = 1; c < C; ++c)
        if (z[c] > m) { m = z[c]; am = c; }
    float s = 0.0f;
    for (int c = 0; c < C; ++c) s += expf(z[c] - m);
    const float lse = logf(s) + m;
    const float zy = (y < C) ? z[y] : 0.0f;
    auto wsum = [&](float v) {
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
        return v;
    };
    const float t0 = wsum(live ? lse - zy : 0.0f), t1 = wsum(live && am == y ? 1.0f : 0.0f);
    if (lane == 0) { ws[wave][0] = t0; ws[wave][1] = t1; }
    for (int c = 0; c < C; ++c) {
        const float pr = expf(z[c] - m) / s;
        const float oh = (c == y) ? 1.0f : 0.0f;
        if (live) dlogits[(size_t)p * C + c] = (pr - oh) * inv_n;
        const float u0 = wsum(live ? oh * pr : 0.0f), u1 = wsum(live ? oh + pr : 0.0f);
        if (lane == 0) { ws[wave][2 + c] = u0; ws[wave][2 + C + c] = u1; }
    }
    __syncthreads();
    if ((int)threadIdx.x < 2 + 2 * C)
        part[(size_t)blockIdx.x * (2 + 2 * C) + threadIdx.x] = ((ws[0][threadIdx.x] + ws[1][threadIdx.x]) + ws[2][threadIdx.x]) + ws[3][threadIdx.x];
}
// acc[slot] = sum over the blocks' rows, strided over the threads in order, then a fixed tree (one workgroup per slot)
__global__ __launch_bounds__(256) void metrics_final_kernel(const float* part, int nblk, int nslot, float* acc) {
    __shared__ float sh[256];
    float s = 0.0f;
    for (int i = threadIdx.x; i < nblk; i += 256) s += part[(size_t)i * nslot + blockIdx.x];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) acc[blockIdx.x] = sh[0];
}

// Gradient of the alternative losses of lib/metrics.py:72-112 with respect to the logits (second pass:
// the per-class sums I_c = sum p_c 1_c and S_c = sum (p_c + 1_c) of the first pass are final).  Keras
// reduces whatever tensor the loss function returns with a mean over all of its elements:
//   dice     L = mean_c -log D_c, D_c = (2 I_c + 100) / (S_c + 100)            (softmax inside)
//   jaccard  L = mean_c -log J_c, J_c = (I_c + 100) / (S_c - I_c + 100)        (softmax inside)
//   dice_and_crossentropy (alpha = 1): dice / 2
//   categorical_hinge on the raw logits: mean_px max(0, max(0, max_{c != y} z_c) - z_y + 1)
//   categorical_focal on the raw logits clipped to [1e-7, 1 - 1e-7] ("y_pred" is what the model outputs):
//            100 * mean over (pixel, class) of -1_c * 0.25 (1 - z_c)^2 log z_c
// acc[2 + 2*PSEG_MAXC] accumulates the hinge / focal loss sum.
__global__ void loss_grad_kernel(int kind, const float* logits, const uint8_t* labels, int n, int C, float inv_n,
                                 const float* acc, float* dlogits, float* alt_sum) {
    __shared__ float sh;
    if (threadIdx.x == 0) sh = 0.0f;
    __syncthreads();
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) {
        const float* z = logits + (size_t)p * C;
        float* dz = dlogits + (size_t)p * C;
        const int y = labels[p];
        if (kind == PSEG_LOSS_DICE || kind == PSEG_LOSS_JACCARD || kind == PSEG_LOSS_DICE_CE) {
            float m = z[0];
            for (int c = 1; c < C; ++c) m = fmaxf(m, z[c]);
            float s = 0.0f, pr[PSEG_MAXC], dp[PSEG_MAXC], dot = 0.0f;
            for (int c = 0; c < C; ++c) { pr[c] = expf(z[c] - m); s += pr[c]; }
            const float scale = (kind == PSEG_LOSS_DICE_CE ? 0.5f : 1.0f) / (float)C;
            for (int c = 0; c < C; ++c) {
                pr[c] /= s;
                const float I = acc[2 + c], S = acc[2 + C + c];
                const float oh = (c == y) ? 1.0f : 0.0f;
                float coef, dcoef;
                if (kind == PSEG_LOSS_JACCARD) {
                    const float den = S - I + 100.0f;
                    coef = (I + 100.0f) / den;
                    dcoef = (oh * den - (I + 100.0f) * (1.0f - oh)) / (den * den);
                } else {
                    const float den = S + 100.0f;
                    coef = (2.0f * I + 100.0f) / den;
                    dcoef = (2.0f * oh * den - (2.0f * I + 100.0f)) / (den * den);
                }
                dp[c] = -scale * dcoef / coef;           // dL/dp_c at this pixel
                dot += dp[c] * pr[c];
            }
            for (int c = 0; c < C; ++c) dz[c] = pr[c] * (dp[c] - dot);      // softmax Jacobian
        } else if (kind == PSEG_LOSS_HINGE) {
            float neg = 0.0f;
            int an = -1;
            for (int c = 0; c < C; ++c)
                if (c != y && z[c] > neg) { neg = z[c]; an = c; }
            const float pos = (y < C) ? z[y] : 0.0f;
            const float l = neg - pos + 1.0f;
            for (int c = 0; c < C; ++c) dz[c] = 0.0f;
            if (l > 0.0f) {
                atomicAdd(&sh, l);
                if (y < C) dz[y] = -inv_n;
                if (an >= 0) dz[an] = inv_n;
            }
        } else {   // focal
            for (int c = 0; c < C; ++c) dz[c] = 0.0f;
            if (y < C) {
                const float eps = 1e-7f;
                const float zc = z[y];
                const float pc = fminf(fmaxf(zc, eps), 1.0f - eps);
                const float k = 100.0f * inv_n / (float)C;
                atomicAdd(&sh, -0.25f * (1.0f - pc) * (1.0f - pc) * logf(pc) * 100.0f / (float)C);
                if (zc > eps && zc < 1.0f - eps)
                    dz[y] = k * 0.25f * (2.0f * (1.0f - pc) * logf(pc) - (1.0f - pc) * (1.0f - pc) / pc);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && sh != 0.0f) atomicAdd(alt_sum, sh);
}

// forward conv weights [KH][KW][Cin][Cout] -> dgrad weights of the channel range [c0, c0+nc):
// Wd[ky'][kx'][co][ci - c0] = W[KH-1-ky'][KW-1-kx'][ci][co]   (+ slack handled by the caller)
__global__ void wd_conv_kernel(const float* w, int KH, int KW, int Cin, int Cout, int c0, int nc, float* wd) {
    const int n = KH * KW * Cout * nc;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int ci = i % nc, co = (i / nc) % Cout, t = i / (nc * Cout);
        const int ky = t / KW, kx = t % KW;
        wd[i] = w[(((size_t)(KH - 1 - ky) * KW + (KW - 1 - kx)) * Cin + c0 + ci) * Cout + co];
    }
    // zero slack behind the kernel: where the matrix-core conv kernels point the lanes that have no weight to load
    if (blockIdx.x == 0 && threadIdx.x < COT) wd[n + threadIdx.x] = 0.0f;
}

// deconv weights [ab][Cin][Cout] -> [ab][Cout][nc] for the channel range [c0, c0+nc)
__global__ void wd_deconv_kernel(const float* w, int Cin, int Cout, int c0, int nc, float* wd) {
    const int n = 4 * Cout * nc;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int ci = i % nc, co = (i / nc) % Cout, ab = i / (nc * Cout);
        wd[i] = w[((size_t)ab * Cin + c0 + ci) * Cout + co];
    }
    if (blockIdx.x == 0 && threadIdx.x < COT) wd[n + threadIdx.x] = 0.0f;   // zero slack (see wd_conv_kernel)
}

// dX[i,j,c] += sum_{ab,co} dY'[2i+a, 2j+b, co] * W[ab][c0+c][co]   (dY' = dY masked by Y > 0 for ReLU)
__global__ __launch_bounds__(256) void deconv2_dgrad_kernel(const float* dY, const float* Y, int relu, int Hin, int Win,
                                                            int Cout, const float* wd /*[ab][Cout][nc]*/, int nc,
                                                            float* dX /*[Hin][Win][nc]*/) {
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= Hin * Win) return;
    const int c0 = blockIdx.y * COT;
    const int i = pix / Win, j = pix - i * Win;
    float acc[COT];
#pragma unroll
    for (int c = 0; c < COT; ++c) acc[c] = 0.0f;
    for (int ab = 0; ab < 4; ++ab) {
        const size_t o = ((size_t)(2 * i + (ab >> 1)) * (2 * Win) + 2 * j + (ab & 1)) * Cout;
        for (int co = 0; co < Cout; ++co) {
            float g = dY[o + co];
            if (relu && !(Y[o + co] > 0.0f)) g = 0.0f;
            const float* wr = wd + ((size_t)ab * Cout + co) * nc + c0;
#pragma unroll
            for (int c = 0; c < COT; ++c) acc[c] = __builtin_fmaf(g, wr[c], acc[c]);
        }
    }
    float* o = dX + (size_t)pix * nc;
#pragma unroll
    for (int c = 0; c < COT; ++c)
        if (c0 + c < nc) o[c0 + c] += acc[c];
}

// max-pool backward: the first maximum of the 2x2 window (row-major) receives the gradient
// fresh != 0: dX holds nothing yet -- all four positions of the window are stored (the gradient at the maximum, zero elsewhere)
__global__ void pool_bwd_kernel(const float* X, const float* dY, int H, int W, int C, float* dX, int fresh) {
    const size_t n = (size_t)(H / 2) * (W / 2) * C;
    const int Wo = W / 2;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(t % C);
        const size_t p = t / C;
        const int x = (int)(p % Wo), y = (int)(p / Wo);
        const size_t b = ((size_t)(2 * y) * W + 2 * x) * C + c;
        const size_t o[4] = {b, b + C, b + (size_t)W * C, b + (size_t)W * C + C};
        int best = 0;
        float bv = X[o[0]];
        for (int q = 1; q < 4; ++q)
            if (X[o[q]] > bv) { bv = X[o[q]]; best = q; }
        if (fresh) {
            const float v = dY[t];
            for (int q = 0; q < 4; ++q) dX[o[q]] = q == best ? v : 0.0f;
        } else {
            dX[o[best]] += dY[t];
        }
    }
}

// dW[tap][ci0 + ci][co] = sum over the strips of part[strip][tap][ci][co] in a FIXED association: groups of WGR_GROUP
// consecutive strips are summed in strip order (wgrad_reduce_groups_kernel, in place into the group's first row: one thread
// per (element, group), so a 15 000-element layer with 340 strips still fills the chip), then the groups in group order.
// dB[co] = the same over the 4 * nstrips bias rows, in float64 (a bias gradient is a sum of a signed map that cancels heavily).
constexpr int WGR_GROUP = 16;
__global__ void wgrad_reduce_groups_kernel(float* part, size_t pstride, int nstrips) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= pstride) return;
    const int k0 = blockIdx.y * WGR_GROUP, k1 = min(k0 + WGR_GROUP, nstrips);
    float s = 0.0f;
    for (int k = k0; k < k1; ++k) s += part[(size_t)k * pstride + e];
    part[(size_t)k0 * pstride + e] = s;
}
__global__ void wgrad_reduce_kernel(const float* part, size_t pstride, int nstrips, int taps, int XC, int Cout, int Cin, int ci0, float* dW,
                                    const float* partB, float* dB) {
    const size_t n = (size_t)taps * XC * Cout;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        float s = 0.0f;
        for (int k = 0; k < nstrips; k += WGR_GROUP) s += part[(size_t)k * pstride + e];
        const int co = (int)(e % Cout);
        const size_t tc = e / Cout;
        const int ci = (int)(tc % XC), tap = (int)(tc / XC);
        dW[((size_t)tap * Cin + ci0 + ci) * Cout + co] = s;
    }
    (void)partB; (void)dB;
}
// out[c] = sum of rows[0 .. nrows)[c] in float64 and a fixed order: one workgroup per channel, the rows strided over its
// threads in order, then a fixed tree (a single thread walking 1 400 rows took 100 us of pure load latency)
template <typename T>
__global__ __launch_bounds__(256) void column_sum_kernel(const T* rows, int nrows, int C, float* out) {
    __shared__ double sh[256];
    const int c = blockIdx.x;
    double s = 0.0;
    for (int k = threadIdx.x; k < nrows; k += 256) s += (double)rows[(size_t)k * C + c];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[c] = (float)sh[0];
}

constexpr int WG_PC = 32;  // pixels per LDS chunk

__global__ __launch_bounds__(256) void wgrad_kernel(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int XCp = (a.XC + 3) & ~3, COp = (a.Cout + 3) & ~3;
    float* Xs = sm;                    // [WG_PC][XCp]
    float* Ys = sm + WG_PC * XCp;      // [WG_PC][COp]
    const int tap = blockIdx.y;
    const int ky = tap / a.KW, kx = tap % a.KW;
    const int itW = a.mode == 0 ? a.Wy : a.Wx, itH = a.mode == 0 ? a.Hy : a.Hx;
    const int r0 = blockIdx.x * a.strip_rows, r1 = min(r0 + a.strip_rows, itH);
    const int tiles_ci = XCp / 4, tiles_co = COp / 4, ntl = tiles_ci * tiles_co;
    float acc[3][16];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[q][i] = 0.0f;
    float bsum = 0.0f;
    const int npx = (r1 - r0) * itW;
    for (int p0 = 0; p0 < npx; p0 += WG_PC) {
        __syncthreads();
        for (int i = threadIdx.x; i < WG_PC * XCp; i += 256) {
            const int pp = i / XCp, c = i - pp * XCp;
            const int p = p0 + pp;
            float v = 0.0f;
            if (p < npx && c < a.XC) {
                const int y = r0 + p / itW, x = p % itW;
                int sy, sx;
                if (a.mode == 0) { sy = y + ky - a.pt; sx = x + kx - a.pl; } else { sy = y; sx = x; }
                if (sy >= 0 && sy < a.Hx && sx >= 0 && sx < a.Wx) v = a.X[((size_t)sy * a.xpitch + sx) * a.XC + c];
            }
            Xs[i] = v;
        }
        for (int i = threadIdx.x; i < WG_PC * COp; i += 256) {
            const int pp = i / COp, c = i - pp * COp;
            const int p = p0 + pp;
            float v = 0.0f;
            if (p < npx && c < a.Cout) {
                const int y = r0 + p / itW, x = p % itW;
                const int dy = a.mode == 0 ? y : 2 * y + (tap >> 1), dx = a.mode == 0 ? x : 2 * x + (tap & 1);
                const size_t o = ((size_t)dy * a.ypitch + dx) * a.Cout + c;
                v = a.dY[o];
                if (a.maskY && !(a.maskY[o] > 0.0f)) v = 0.0f;
            }
            Ys[i] = v;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int tl = threadIdx.x + q * 256;
            if (tl < ntl) {
                const int tci = tl / tiles_co, tco = tl - tci * tiles_co;
                for (int pp = 0; pp < WG_PC; ++pp) {
                    const float4 xv = *(const float4*)(Xs + pp * XCp + tci * 4);
                    const float4 yv = *(const float4*)(Ys + pp * COp + tco * 4);
                    const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ys[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[q][i * 4 + j] = __builtin_fmaf(xs[i], ys[j], acc[q][i * 4 + j]);
                }
            }
        }
        if (a.dB && tap == 0 && (int)threadIdx.x < a.Cout)
            for (int pp = 0; pp < WG_PC; ++pp) bsum += Ys[pp * COp + threadIdx.x];
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int tl = threadIdx.x + q * 256;
        if (tl < ntl) {
            const int tci = tl / tiles_co, tco = tl - tci * tiles_co;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ci = tci * 4 + i, co = tco * 4 + j;
                    if (ci < a.XC && co < a.Cout) wg_out(a, blockIdx.x, tap, ci, co, acc[q][i * 4 + j]);
                }
        }
    }
    if (a.dB && tap == 0 && (int)threadIdx.x < a.Cout)
        for (int w = 0; w < 4; ++w) wg_out_bias(a, blockIdx.x, w, threadIdx.x, w == 0 ? bsum : 0.0f);
}

// Matrix-core weight gradient.  dW[tap][ci][co] += sum over pixels of X[pixel + tap][ci] * dY[pixel][co]
// runs as v_mfma_f32_16x16x4_f32 with the pixel index as the MFMA k dimension: A = X^T (rows = 16
// input channels, k = 4 pixels), B = dY (k = 4 pixels, columns = 16 output channels).  One
// workgroup = one tap x one strip of rows; every wave walks its own pixel quads of the strip and
// loads both operands straight from global memory in fragment layout (lane = (channel, pixel):
// four 64-byte segments per instruction; nothing is shared between waves, so no LDS staging and
// no barriers in the loop) for ALL (ci, co) tiles (<= 5 x 5 accumulator tiles = 100 VGPRs).  The
// tap is the fastest grid dimension: the workgroups of one strip run together and re-read it
// from L2.  The four waves' partial sums are reduced through LDS and leave as one atomicAdd per
// element.  Accumulation order is free here (float tolerance against torch autograd, not bit parity).
typedef __attribute__((ext_vector_type(4))) float wg_f32x4;
constexpr int WGM_MAXT = 5;   // up to 80 channels per side

// TI x TJ = accumulator tiles (input-channel tiles x output-channel tiles) the instance keeps: small
// layers get small instances, hence many waves per SIMD to hide the operand-load latency.
// KXN = kernel columns handled by one workgroup (1, or the whole kernel row of a k5 layer when
// 5 x TI x TJ accumulator tiles fit): the dY fragments of a pixel quad are then loaded once for the
// five taps of the row instead of five times.
template <int TI, int TJ, int KXN>
__global__ __launch_bounds__(256) void wgrad_mfma_kernel(WgradArgs a) {
    __shared__ float red[256];
    const int tap0 = blockIdx.x * KXN;                 // first tap of this workgroup
    const int ky = tap0 / a.KW, kx0 = tap0 % a.KW;
    const int itW = a.mode == 0 ? a.Wy : a.Wx, itH = a.mode == 0 ? a.Hy : a.Hx;
    const int r0 = blockIdx.y * a.strip_rows, r1 = min(r0 + a.strip_rows, itH);
    // channel blocks of TI x TJ tiles (blockIdx.z) for layers wider than one instance
    const int nbo = (a.Cout + TJ * 16 - 1) / (TJ * 16);
    const int xc0 = ((int)blockIdx.z / nbo) * TI * 16, yc0 = ((int)blockIdx.z % nbo) * TJ * 16;
    const int XCb = min(a.XC - xc0, TI * 16), COb = min(a.Cout - yc0, TJ * 16);
    const int tiles_ci = (XCb + 15) >> 4, tiles_co = (COb + 15) >> 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, p16 = lane & 15, g = lane >> 4;
    wg_f32x4 acc[KXN][TI][TJ];
#pragma unroll
    for (int k = 0; k < KXN; ++k)
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) acc[k][i][j] = wg_f32x4{0.f, 0.f, 0.f, 0.f};
    float bacc[TJ];
#pragma unroll
    for (int j = 0; j < TJ; ++j) bacc[j] = 0.0f;
    const bool want_b = a.dB != nullptr && tap0 == 0 && xc0 == 0;

    // (fetched values are not looked at before the rotation at the end of the step: a ReLU or mask select right behind its
    // load put a full memory wait behind every dY fragment -- five to six serial latencies per quad of pixels)
    auto load = [&](int y, int xq, float (*xa)[TI], float* yb, float* ym) {
        const int x = xq + g;
#pragma unroll
        for (int k = 0; k < KXN; ++k) {
            int sy, sx;
            if (a.mode == 0) { sy = y * a.stride + ky - a.pt; sx = x * a.stride + kx0 + k - a.pl; } else { sy = y; sx = x; }
            const bool okx = x < itW && sy >= 0 && sy < a.Hx && sx >= 0 && sx < a.Wx;
            const float* xp = a.X + ((size_t)(sy >> a.xup) * a.xpitch + (sx >> a.xup)) * a.XC + xc0 + p16;
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                xa[k][i] = 0.0f;
                if (i < tiles_ci && okx && i * 16 + p16 < XCb) xa[k][i] = xp[i * 16];
            }
        }
        const int dy = a.mode == 0 ? y : 2 * y + (tap0 >> 1), dx = a.mode == 0 ? x : 2 * x + (tap0 & 1);
        const size_t o = ((size_t)dy * a.ypitch + dx) * a.Cout + yc0 + p16;
        const bool oky = x < itW;
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            yb[j] = 0.0f;
            ym[j] = 1.0f;
            if (j < tiles_co && oky && j * 16 + p16 < COb) {
                yb[j] = a.dY[o + j * 16];
                if (a.maskY) ym[j] = a.maskY[o + j * 16];
            }
        }
    };
    // quads of this wave: (row y, columns 4*(wave + 4n) .. +3), walked as one sequence
    const int qpr = (itW + 15) >> 4;                   // quads per wave per row
    const int nq = (r1 - r0) * qpr;
    float xa[KXN][TI], yb[TJ], xn[KXN][TI], yn[TJ], ymn[TJ];
    auto rotate = [&]() {
#pragma unroll
        for (int k = 0; k < KXN; ++k)
#pragma unroll
            for (int i = 0; i < TI; ++i) xa[k][i] = (a.in_relu && !(xn[k][i] > 0.0f)) ? 0.0f : xn[k][i];
#pragma unroll
        for (int j = 0; j < TJ; ++j) yb[j] = ymn[j] > 0.0f ? yn[j] : 0.0f;
    };
    if (nq > 0) { load(r0, wave * 4, xn, yn, ymn); rotate(); }
    for (int q = 0; q < nq; ++q) {
        if (q + 1 < nq) {
            const int qn = q + 1, yr = qn / qpr, xc = qn - yr * qpr;
            load(r0 + yr, (xc * 4 + wave) * 4, xn, yn, ymn);
        }
#pragma unroll
        for (int k = 0; k < KXN; ++k)
#pragma unroll
            for (int i = 0; i < TI; ++i)
                if (i < tiles_ci)
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
                        if (j < tiles_co) acc[k][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[k][i], yb[j], acc[k][i][j], 0, 0, 0);
        if (want_b)
#pragma unroll
            for (int j = 0; j < TJ; ++j) bacc[j] += yb[j];
        if (q + 1 < nq) rotate();
    }
    // D tile: lane holds rows (ci) 4g..4g+3, column (co) p16.  Reduce the four waves through LDS.
#pragma unroll
    for (int k = 0; k < KXN; ++k)
#pragma unroll
    for (int i = 0; i < TI; ++i)
        if (i < tiles_ci)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
                if (j < tiles_co) {
                    for (int w = 0; w < 4; ++w) {
                        if (wave == w) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                float* qd = red + (4 * g + r) * 16 + p16;
                                *qd = w == 0 ? acc[k][i][j][r] : *qd + acc[k][i][j][r];
                            }
                        }
                        __syncthreads();
                    }
                    {
                        const int e = threadIdx.x, row = e >> 4, col = e & 15;
                        const int ci = i * 16 + row, co = j * 16 + col;
                        if (ci < XCb && co < COb) wg_out(a, blockIdx.y, tap0 + k, xc0 + ci, yc0 + co, red[e]);
                    }
                    __syncthreads();
                }
    if (want_b) {
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            float v = bacc[j];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (g == 0 && j < tiles_co && j * 16 + p16 < COb) wg_out_bias(a, blockIdx.y, wave, yc0 + j * 16 + p16, v);
        }
    }
}

// LDS-staged form of wgrad_mfma_kernel for stride-1 convolutions (every fcn / fcn_skip layer, unet's convs): the same
// MFMA formulation (k = pixel, A = X^T, B = dY), but a workgroup brings a 64-pixel piece of the X row (+ the KXN - 1
// columns the kernel row reaches) and of the dY row into LDS with coalesced loads -- ReLU mask and pre-activation ReLU
// applied on the way -- laid out [16-channel tile][pixel][16], so that a fragment is one conflict-free ds_read_b32 per lane
// (lane (c, g) reads pixel q + g (+ kx), channel c).  The next piece is fetched into registers while the current one is
// multiplied, which takes the global-memory latency off the MFMA path: the direct kernel issued twelve 4-byte gathers
// per quad of pixels and ran the matrix pipe at a quarter of its rate.
constexpr int WGL_CW = 64;   // pixels of a row per staged piece
template <int TI, int TJ, int KXN>
__global__ __launch_bounds__(256) void wgrad_lds_kernel(WgradArgs a) {
    constexpr int NXP = WGL_CW + KXN - 1;
    constexpr int EX = (NXP * TI * 16 + 255) / 256, EY = (WGL_CW * TJ * 16 + 255) / 256;
    __shared__ float Xs[TI * NXP * 16];
    __shared__ float Ys[TJ * WGL_CW * 16];
    __shared__ float red[256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p16 = lane & 15, g = lane >> 4;
    const int tap0 = blockIdx.x * KXN;
    const int ky = tap0 / a.KW, kx0 = tap0 % a.KW;
    const int r0 = blockIdx.y * a.strip_rows, r1 = min(r0 + a.strip_rows, a.Hy);
    const int nbo = (a.Cout + TJ * 16 - 1) / (TJ * 16);
    const int xc0 = ((int)blockIdx.z / nbo) * TI * 16, yc0 = ((int)blockIdx.z % nbo) * TJ * 16;
    const int XCb = min(a.XC - xc0, TI * 16), COb = min(a.Cout - yc0, TJ * 16);
    const int tiles_ci = (XCb + 15) >> 4, tiles_co = (COb + 15) >> 4;
    const unsigned invx = (1u << 20) / (unsigned)XCb + 1u, invy = (1u << 20) / (unsigned)COb + 1u;
    const int NX = NXP * XCb, NY = WGL_CW * COb;
    wg_f32x4 acc[KXN][TI][TJ];
#pragma unroll
    for (int k = 0; k < KXN; ++k)
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) acc[k][i][j] = wg_f32x4{0.f, 0.f, 0.f, 0.f};
    float bacc[TJ];
#pragma unroll
    for (int j = 0; j < TJ; ++j) bacc[j] = 0.0f;
    const bool want_b = a.dB != nullptr && tap0 == 0 && xc0 == 0;
    // zero the channel pad of the tiles once (staging only writes channels < XCb / COb)
    for (int i = tid; i < TI * NXP * 16; i += 256) Xs[i] = 0.0f;
    for (int i = tid; i < TJ * WGL_CW * 16; i += 256) Ys[i] = 0.0f;
    const int cpr = (a.Wy + WGL_CW - 1) / WGL_CW;      // pieces per row
    const int npieces = (r1 - r0) * cpr;
    float xr[EX], yr[EY], ym[EY];
    // element e = tid + 256 u of a piece is pixel e / channels, channel e % channels for every piece: taken apart once,
    // kept packed (pixel << 8 | channel; 0xFFFF00 = past the piece)
    int ex[EX], ey[EY];
#pragma unroll
    for (int u = 0; u < EX; ++u) {
        const int e = tid + u * 256;
        const int px = (int)(((unsigned)e * invx) >> 20), c = e - px * XCb;
        ex[u] = e < NX ? (px << 8 | c) : 0xFFFF00;
    }
#pragma unroll
    for (int u = 0; u < EY; ++u) {
        const int e = tid + u * 256;
        const int px = (int)(((unsigned)e * invy) >> 20), c = e - px * COb;
        ey[u] = e < NY ? (px << 8 | c) : 0xFFFF00;
    }
    auto fetch = [&](int piece) {
        const int row = piece / cpr, y = r0 + row, x0 = (piece - row * cpr) * WGL_CW;
        const int sy = y + ky - a.pt, sxb = x0 + kx0 - a.pl;
        const bool rowok = sy >= 0 && sy < a.Hx;
        const float* xrow = a.X + ((size_t)(rowok ? sy : 0) * a.xpitch + sxb) * a.XC + xc0;   // (never dereferenced out of range)
#pragma unroll
        for (int u = 0; u < EX; ++u) {
            const int px = ex[u] >> 8, c = ex[u] & 255;
            // (nothing looks at a fetched value before commit(): a use right behind a load would put a full memory wait
            // behind every one of them -- the first version of this kernel did, and ran six times under its MFMA time)
            xr[u] = 0.0f;
            if (rowok && (unsigned)(sxb + px) < (unsigned)a.Wx) xr[u] = xrow[px * a.XC + c];
        }
        const float* yrow = a.dY + ((size_t)y * a.ypitch + x0) * a.Cout + yc0;
        const float* mrow = a.maskY ? a.maskY + ((size_t)y * a.ypitch + x0) * a.Cout + yc0 : nullptr;
#pragma unroll
        for (int u = 0; u < EY; ++u) {
            const int px = ey[u] >> 8, c = ey[u] & 255;
            yr[u] = 0.0f;
            ym[u] = 1.0f;
            if (x0 + px < a.Wy) {
                yr[u] = yrow[px * a.Cout + c];
                if (mrow) ym[u] = mrow[px * a.Cout + c];
            }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int u = 0; u < EX; ++u) {
            const int px = ex[u] >> 8, c = ex[u] & 255;
            if (px < NXP) Xs[((c >> 4) * NXP + px) * 16 + (c & 15)] = (a.in_relu && !(xr[u] > 0.0f)) ? 0.0f : xr[u];
        }
#pragma unroll
        for (int u = 0; u < EY; ++u) {
            const int px = ey[u] >> 8, c = ey[u] & 255;
            if (px < WGL_CW) Ys[((c >> 4) * WGL_CW + px) * 16 + (c & 15)] = ym[u] > 0.0f ? yr[u] : 0.0f;
        }
    };
    if (npieces > 0) fetch(0);
    __syncthreads();                                   // the zero fill is done
    for (int piece = 0; piece < npieces; ++piece) {
        commit();
        __syncthreads();
        if (piece + 1 < npieces) fetch(piece + 1);     // in flight under the MFMAs below
#pragma unroll
        for (int q4 = 0; q4 < WGL_CW / 16; ++q4) {
            const int xq = (q4 * 4 + wave) * 4 + g;
            float xa[KXN][TI], yb[TJ];
#pragma unroll
            for (int j = 0; j < TJ; ++j) yb[j] = Ys[(j * WGL_CW + xq) * 16 + p16];
#pragma unroll
            for (int k = 0; k < KXN; ++k)
#pragma unroll
                for (int i = 0; i < TI; ++i) xa[k][i] = Xs[(i * NXP + xq + k) * 16 + p16];
#pragma unroll
            for (int k = 0; k < KXN; ++k)
#pragma unroll
                for (int i = 0; i < TI; ++i)
                    if (i < tiles_ci)
#pragma unroll
                        for (int j = 0; j < TJ; ++j)
                            if (j < tiles_co) acc[k][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[k][i], yb[j], acc[k][i][j], 0, 0, 0);
            if (want_b)
#pragma unroll
                for (int j = 0; j < TJ; ++j) bacc[j] += yb[j];
        }
        __syncthreads();                               // every wave is done with this piece
    }
#pragma unroll
    for (int k = 0; k < KXN; ++k)
#pragma unroll
    for (int i = 0; i < TI; ++i)
        if (i < tiles_ci)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
                if (j < tiles_co) {
                    for (int w = 0; w < 4; ++w) {
                        if (wave == w) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                float* qd = red + (4 * g + r) * 16 + p16;
                                *qd = w == 0 ? acc[k][i][j][r] : *qd + acc[k][i][j][r];
                            }
                        }
                        __syncthreads();
                    }
                    {
                        const int e = threadIdx.x, row = e >> 4, col = e & 15;
                        const int ci = i * 16 + row, co = j * 16 + col;
                        if (ci < XCb && co < COb) wg_out(a, blockIdx.y, tap0 + k, xc0 + ci, yc0 + co, red[e]);
                    }
                    __syncthreads();
                }
    if (want_b) {
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            float v = bacc[j];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (g == 0 && j < tiles_co && j * 16 + p16 < COb) wg_out_bias(a, blockIdx.y, wave, yc0 + j * 16 + p16, v);
        }
    }
}

// Weight gradient of a ONE-channel input layer (the first conv: Cin = 1).  With the input channel as the MFMA row the
// instance above fills one of its sixteen rows; here the rows are the KW x KW taps instead: A[tap][pixel] = X[pixel + tap]
// (an im2col gather through L1, 4 bytes per lane), B[pixel][cout] = dY, so one 16x16x4 MFMA covers sixteen taps of four
// pixels.  One workgroup = a strip of rows, every wave walks its own pixel quads; partial sums leave through the same
// LDS reduction + one atomicAdd per element as wgrad_mfma_kernel.  (2048x1536, 1 -> 20, k5: 2.28 ms -> see DESIGN 5.)
template <int TM, int TJ>
__global__ __launch_bounds__(256) void wgrad_c1_kernel(WgradArgs a) {
    __shared__ float red[256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, p16 = lane & 15, g = lane >> 4;
    const int KK = a.KW * a.KW;
    const int r0 = blockIdx.x * a.strip_rows, r1 = min(r0 + a.strip_rows, a.Hy);
    const int tiles_co = (a.Cout + 15) >> 4;
    int tky[TM], tkx[TM];
    bool tok[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int tap = i * 16 + p16;
        tok[i] = tap < KK;
        tky[i] = (tok[i] ? tap / a.KW : 0) - a.pt;
        tkx[i] = (tok[i] ? tap % a.KW : 0) - a.pl;
    }
    wg_f32x4 acc[TM][TJ];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = wg_f32x4{0.f, 0.f, 0.f, 0.f};
    float bacc[TJ];
#pragma unroll
    for (int j = 0; j < TJ; ++j) bacc[j] = 0.0f;
    const bool want_b = a.dB != nullptr;
    auto load = [&](int y, int xq, float* xa, float* yb) {
        const int x = xq + g;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int sy = y + tky[i], sx = x + tkx[i];
            const bool ok = tok[i] && x < a.Wy && sy >= 0 && sy < a.Hx && sx >= 0 && sx < a.Wx;
            xa[i] = ok ? a.X[(size_t)sy * a.xpitch + sx] : 0.0f;
        }
        const size_t o = ((size_t)y * a.ypitch + x) * a.Cout + p16;
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            float v = 0.0f;
            if (j < tiles_co && x < a.Wy && j * 16 + p16 < a.Cout) {
                v = a.dY[o + j * 16];
                if (a.maskY && !(a.maskY[o + j * 16] > 0.0f)) v = 0.0f;
            }
            yb[j] = v;
        }
    };
    const int qpr = (a.Wy + 15) >> 4;                  // quads per wave per row
    const int nq = (r1 - r0) * qpr;
    float xa[TM], yb[TJ], xn[TM], yn[TJ];
    if (nq > 0) load(r0, wave * 4, xa, yb);
    for (int q = 0; q < nq; ++q) {
        if (q + 1 < nq) {
            const int qn = q + 1, yr = qn / qpr, xc = qn - yr * qpr;
            load(r0 + yr, (xc * 4 + wave) * 4, xn, yn);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
                if (j < tiles_co) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[i], yb[j], acc[i][j], 0, 0, 0);
        if (want_b)
#pragma unroll
            for (int j = 0; j < TJ; ++j) bacc[j] += yb[j];
#pragma unroll
        for (int i = 0; i < TM; ++i) xa[i] = xn[i];
#pragma unroll
        for (int j = 0; j < TJ; ++j) yb[j] = yn[j];
    }
    // D tile: lane holds rows (taps) 4g..4g+3, column (cout) p16.  Reduce the four waves through LDS.
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
            if (j < tiles_co) {
                for (int w = 0; w < 4; ++w) {
                    if (wave == w) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float* qd = red + (4 * g + r) * 16 + p16;
                            *qd = w == 0 ? acc[i][j][r] : *qd + acc[i][j][r];
                        }
                    }
                    __syncthreads();
                }
                {
                    const int e = threadIdx.x, row = e >> 4, col = e & 15;
                    const int tap = i * 16 + row, co = j * 16 + col;
                    if (tap < KK && co < a.Cout) wg_out(a, blockIdx.x, tap, 0, co, red[e]);
                }
                __syncthreads();
            }
    if (want_b) {
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            float v = bacc[j];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (g == 0 && j < tiles_co && j * 16 + p16 < a.Cout) wg_out_bias(a, blockIdx.x, wave, j * 16 + p16, v);
        }
    }
}

// Chooses the kernel instance for a layer, launches it and (deterministic form) sums the strips' partial results.
// grid = (strips, taps) as the scalar kernel takes it; `scratch` / `scratch_cap` = the train state's partial-sum buffer.
static int launch_wgrad(const WgradArgs& a_in, dim3 grid, hipStream_t st, float** scratch, size_t* scratch_cap) {
    WgradArgs a = a_in;
    const int itH = a.mode == 0 ? a.Hy : a.Hx;
    const int taps = (int)grid.y;
    // ---- which instance, with which strips
    enum { V_PAIR, V_FLAT, V_C1, V_KX5, V_TAP, V_WIDE, V_SCALAR } variant;
    const bool scalar = false;
    WgradFlatPlan flat;
    const bool lds = a.mode == 0 && a.stride == 1 && !a.xup;
    const int ti = (a.XC + 15) / 16, tj = (a.Cout + 15) / 16;
    if (a.XC0 > 0) {                                          // both concat sources + the bias gradient in one pass (pseg_wgrad_flat.hip)
        if (!wgrad_pair_plan(a, taps, &flat)) return fail(PSEG_EINVAL, "no two-source weight-gradient instance for this layer");
        variant = V_PAIR;
        a.strip_rows = flat.strip_rows;
    } else if (!scalar && wgrad_flat_plan(a, taps, &flat)) {
        variant = V_FLAT;                                      // the k5 layers of fcn / fcn_skip (pseg_wgrad_flat.hip)
        a.strip_rows = flat.strip_rows;
    } else if (a.XC == 1 && a.mode == 0 && a.stride == 1 && !a.xup && !a.in_relu && a.KW * a.KW <= 32 && a.Cout <= 64 && !scalar) {
        variant = V_C1;
        a.strip_rows = std::max(1, cdiv(a.Hy, 1024));
    } else if (a.XC <= 16 * WGM_MAXT && a.Cout <= 16 * WGM_MAXT && !scalar) {
        // a whole kernel row per workgroup for the small k5 layers (mode 0: taps of a row share dY)
        if (a.mode == 0 && a.KW == 5 && taps % 5 == 0 && ((ti <= 2 && tj <= 2) || (ti <= 2 && tj <= 3) || (ti <= 1 && tj <= 2))) {
            variant = V_KX5;                                   // five times fewer "taps": five times more strips
            a.strip_rows = std::max(1, a.strip_rows / 5);
        } else {
            variant = V_TAP;
        }
    } else if (!scalar) {
        variant = V_WIDE;
        // wide layers sit on small maps and get their parallelism from the 64 x 64 channel blocks and the taps: few strips
        // (every strip is a full copy of the layer's gradient in the partial-sum buffer -- 19 MB for 512 -> 1024 channels; at one
        // row per strip the reduction of unet's partials alone took 2 ms of a 25 ms step)
        const int nblk = cdiv(a.XC, 64) * cdiv(a.Cout, 64);
        const int want = std::max(1, 2048 / std::max(1, taps * nblk));
        a.strip_rows = std::max(a.strip_rows, cdiv(itH, want));
    } else {
        variant = V_SCALAR;
    }
    const int nstrips = (variant == V_FLAT || variant == V_PAIR) ? flat.nstrips : cdiv(itH, a.strip_rows);
    // ---- deterministic form: partial sums per strip, then one ordered reduction (PSEG_WGRAD_ATOMIC=1: float atomics)
    const bool det = !PSEG_KNOB("PSEG_WGRAD_ATOMIC");
    if (det) {
        a.pstride = (size_t)taps * a.XC * a.Cout;
        const size_t need = ((size_t)nstrips * a.pstride + (size_t)4 * nstrips * a.Cout) * sizeof(float);
        if (*scratch_cap < need) {
            if (*scratch) { PSEG_HIP(hipStreamSynchronize(st)); (void)hipFree(*scratch); }
            *scratch = nullptr; *scratch_cap = 0;
            PSEG_HIP(hipMalloc((void**)scratch, need));
            *scratch_cap = need;
        }
        a.part = *scratch;
        a.partB = a.dB ? *scratch + (size_t)nstrips * a.pstride : nullptr;
    }
    switch (variant) {
        case V_PAIR: PSEG_TRY(wgrad_pair_launch(a, flat, st)); break;
        case V_FLAT: PSEG_TRY(wgrad_flat_launch(a, flat, st)); break;
        case V_C1: {
            if (a.KW * a.KW <= 16) {
                if (tj <= 2) wgrad_c1_kernel<1, 2><<<nstrips, 256, 0, st>>>(a);
                else wgrad_c1_kernel<1, 4><<<nstrips, 256, 0, st>>>(a);
            } else {
                if (tj <= 2) wgrad_c1_kernel<2, 2><<<nstrips, 256, 0, st>>>(a);
                else wgrad_c1_kernel<2, 4><<<nstrips, 256, 0, st>>>(a);
            }
            break;
        }
        case V_KX5: {
            const dim3 g5(taps / 5, nstrips);
#define PSEG_WG5(TI_, TJ_) if (ti <= TI_ && tj <= TJ_) {                                                   \
                if (lds) wgrad_lds_kernel<TI_, TJ_, 5><<<g5, 256, 0, st>>>(a);                                       \
                else wgrad_mfma_kernel<TI_, TJ_, 5><<<g5, 256, 0, st>>>(a);                                          \
                break; }
            PSEG_WG5(1, 2) PSEG_WG5(2, 2) PSEG_WG5(2, 3)
#undef PSEG_WG5
            // (a kernel row per workgroup for the wider layers -- <3,3,5>, <3,4,5>, <4,3,5> on the LDS-staged kernel, 384-497
            // registers, one wave per SIMD -- was measured: 30.17 vs 29.97 ms per step against one tap per workgroup below)
            return fail(PSEG_EUNSUPPORTED, "no kernel-row weight-gradient instance for %d x %d channels", a.XC, a.Cout);
        }
        case V_TAP: {
            const dim3 g2(taps, nstrips);
#define PSEG_WG(TI_, TJ_) if (ti <= TI_ && tj <= TJ_) {                                                    \
            wgrad_mfma_kernel<TI_, TJ_, 1><<<g2, 256, 0, st>>>(a);   /* one tap per workgroup: the direct kernel is faster than the LDS-staged one */ \
            break; }
            PSEG_WG(1, 2) PSEG_WG(2, 2) PSEG_WG(2, 3) PSEG_WG(3, 3) PSEG_WG(3, 5) PSEG_WG(5, 3) PSEG_WG(5, 5)
#undef PSEG_WG
            return fail(PSEG_EUNSUPPORTED, "no weight-gradient instance for %d x %d channels", a.XC, a.Cout);
        }
        case V_WIDE: {
            // wide layers (unet / res_unet): 64 x 64 channel blocks on blockIdx.z
            const dim3 g3(taps, nstrips, cdiv(a.XC, 64) * cdiv(a.Cout, 64));
            wgrad_mfma_kernel<4, 4, 1><<<g3, 256, 0, st>>>(a);
            break;
        }
        case V_SCALAR: {
            const int XCp = (a.XC + 3) & ~3, COp = (a.Cout + 3) & ~3;
            if (a.stride != 1 || a.xup || a.in_relu || (XCp / 4) * (COp / 4) > 768)
                return fail(PSEG_EUNSUPPORTED, "the scalar weight-gradient kernel does not cover this layer");
            wgrad_kernel<<<dim3(nstrips, taps), 256, (size_t)WG_PC * (XCp + COp) * 4, st>>>(a);
            break;
        }
    }
    PSEG_HIP(hipGetLastError());
    if (det) {
        const size_t n = a.pstride;
        wgrad_reduce_groups_kernel<<<dim3((unsigned)((n + 255) / 256), (unsigned)cdiv(nstrips, WGR_GROUP)), 256, 0, st>>>(a.part, a.pstride, nstrips);
        wgrad_reduce_kernel<<<(int)std::min<size_t>((n + 255) / 256, 2048), 256, 0, st>>>(a.part, a.pstride, nstrips, taps, a.XC, a.Cout, a.Cin, a.ci0, a.dW,
                                                                                          a.partB, a.dB);
        if (a.partB) column_sum_kernel<float><<<a.Cout, 256, 0, st>>>(a.partB, 4 * nstrips, a.Cout, a.dB);
        PSEG_HIP(hipGetLastError());
    }
    return PSEG_OK;
}

// dst[i] += src[i] (masked by maskX[i] > 0 when given): residual addend gradients, pre-activation ReLU dgrad
__global__ void accum_kernel(float* dst, const float* src, const float* maskY, const float* maskX, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float v = src[i];
        if (maskY && !(maskY[i] > 0.0f)) v = 0.0f;
        if (maskX && !(maskX[i] > 0.0f)) v = 0.0f;
        dst[i] += v;
    }
}
// gradient of a nearest x2 upsample: dst[Y][X][c] += sum of the 2x2 block of src (2H x 2W x C), masked by maskX
__global__ void upsample_bwd_kernel(float* dst, const float* src, const float* maskX, int H, int W, int C) {
    const size_t n = (size_t)H * W * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const size_t p = i / C;
        const int x = (int)(p % W), y = (int)(p / W);
        const float* q = src + ((size_t)(2 * y) * (2 * W) + 2 * x) * C + c;
        float v = (q[0] + q[C]) + (q[(size_t)2 * W * C] + q[(size_t)2 * W * C + C]);
        if (maskX && !(maskX[i] > 0.0f)) v = 0.0f;
        dst[i] += v;
    }
}
// zero-dilated copy of a (ReLU-masked) gradient for the data gradient of a stride-2 conv: dst (2H x 2W x C), dst(2y, 2x) = src(y, x)
__global__ void dilate2_kernel(float* dst, const float* src, const float* maskY, int H, int W, int C) {
    const size_t n = (size_t)4 * H * W * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const size_t p = i / C;
        const int x = (int)(p % (2 * W)), y = (int)(p / (2 * W));
        float v = 0.0f;
        if (!(x & 1) && !(y & 1)) {
            const size_t o = ((size_t)(y >> 1) * W + (x >> 1)) * C + c;
            v = src[o];
            if (maskY && !(maskY[o] > 0.0f)) v = 0.0f;
        }
        dst[i] = v;
    }
}

// dB[co] = sum over pixels of dY' (dY masked by Y > 0 for ReLU layers), in a fixed order: a thread owns one channel of every
// (256 / Cout)-th pixel of its block's stripe and sums it in float64 (a bias gradient is a sum of a signed map over up to 3.1 M
// pixels that cancels heavily), the block's threads of a channel are added in thread order, the blocks in block order.
constexpr int BG_BLOCKS = 512;
__global__ __launch_bounds__(256) void bias_grad_kernel(const float* dY, const float* maskY, size_t npix, int Cout, double* part) {
    __shared__ double sh[256];
    const int ppb = 256 / Cout;                        // pixels per block trip
    const int c = (int)threadIdx.x % Cout, pg = (int)threadIdx.x / Cout;
    double s = 0.0;
    if (pg < ppb)
        for (size_t px = (size_t)blockIdx.x * ppb + pg; px < npix; px += (size_t)gridDim.x * ppb) {
            const size_t i = px * Cout + c;
            float v = dY[i];
            if (maskY && !(maskY[i] > 0.0f)) v = 0.0f;
            s += (double)v;
        }
    sh[threadIdx.x] = s;
    __syncthreads();
    if ((int)threadIdx.x < Cout) {
        double tsum = 0.0;
        for (int k = 0; k < ppb; ++k) tsum += sh[k * Cout + threadIdx.x];
        part[(size_t)blockIdx.x * Cout + threadIdx.x] = tsum;
    }
}
// Sum of squares of one parameter's (scaled) gradient, in a FIXED summation order: per-thread stripes, butterfly inside the
// wave, the four waves in order through LDS, then the blocks' partial sums in order (sumsq_final_kernel).  Data-parallel
// replicas clip by this norm: with float atomics its last bit depended on the arrival order and two ranks that hold the
// same all-reduced gradient could step apart (tests/test_configs_gpu.py: replicas bit-identical).
constexpr int SUMSQ_MAXB = 1024;
__global__ __launch_bounds__(256) void sumsq_kernel(const float* g, int64_t n, float scale, float* part) {
    __shared__ float ws[4];
    float s = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = g[i] * scale;
        s += v * v;
    }
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = ((ws[0] + ws[1]) + ws[2]) + ws[3];
}
// one block per (parameter slice): its nblk partial sums, strided over the threads in order, tree-reduced in LDS; slices
// of one tensor (a BatchNormalization vector spread over two ops: exactly two terms) meet in out[pi] -- a + b == b + a
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* part, const int* nblk, const int* pidx, float* out) {
    __shared__ float sh[256];
    const float* p = part + (size_t)blockIdx.x * SUMSQ_MAXB;
    const int nb = nblk[blockIdx.x];
    float s = 0.0f;
    for (int i = threadIdx.x; i < nb; i += 256) s += p[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(out + pidx[blockIdx.x], sh[0]);
}

// g <- g*scale; per-tensor clip_by_norm (t * clip / max(norm, clip)); optional clipvalue; then the update
// rule of the chosen Keras (TF 2.5 optimizer_v2) optimizer with its default hyper-parameters -- the
// reference only ever passes lr / clipnorm / clipvalue (lib/network.py:92-102):
//   adam     m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= lr sqrt(1-b2^t)/(1-b1^t) m / (sqrt(v) + eps)
//   sgd      p -= lr g
//   rmsprop  v = .9 v + .1 g^2; p -= lr g / sqrt(v + eps)                      (fused ApplyRMSProp form)
//   adagrad  v += g^2 (v0 = 0.1); p -= lr g / (sqrt(v) + eps)
//   adadelta v = .95 v + .05 g^2; u = sqrt(m + eps) / sqrt(v + eps) g; p -= lr u; m = .95 m + .05 u^2
//   adamax   m = b1 m + (1-b1) g; v = max(b2 v, |g|); p -= lr/(1-b1^t) m / (v + eps)
//   nadam    Keras' momentum-schedule form; the four schedule scalars c0..c3 come from the host
struct OptScalars { float lr, b1, b2, eps, inv_ms_new, inv_ms_next, inv_b2t, u_t, u_t1; };

__global__ void opt_kernel(int kind, float* p, const float* g, float* m, float* v, int64_t n, float gscale, const float* sumsq,
                           float clipnorm, float clipvalue, OptScalars o) {
    float cn = 1.0f;
    if (clipnorm > 0.0f) {
        const float norm = sqrtf(*sumsq);
        cn = clipnorm / fmaxf(norm, clipnorm);
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float gi = g[i] * gscale * cn;
        if (clipvalue > 0.0f) gi = fminf(fmaxf(gi, -clipvalue), clipvalue);
        switch (kind) {
            case PSEG_OPT_ADAM: {
                const float mi = o.b1 * m[i] + (1.0f - o.b1) * gi;
                const float vi = o.b2 * v[i] + (1.0f - o.b2) * gi * gi;
                m[i] = mi; v[i] = vi;
                p[i] -= o.lr * mi / (sqrtf(vi) + o.eps);      // o.lr = lr_t
            } break;
            case PSEG_OPT_SGD: p[i] -= o.lr * gi; break;
            case PSEG_OPT_RMSPROP: {
                const float vi = v[i] + (gi * gi - v[i]) * (1.0f - 0.9f);
                v[i] = vi;
                p[i] -= o.lr * gi / sqrtf(vi + o.eps);
            } break;
            case PSEG_OPT_ADAGRAD: {
                const float vi = v[i] + gi * gi;
                v[i] = vi;
                p[i] -= o.lr * gi / (sqrtf(vi) + o.eps);
            } break;
            case PSEG_OPT_ADADELTA: {
                const float vi = 0.95f * v[i] + 0.05f * gi * gi;
                const float u = sqrtf(m[i] + o.eps) / sqrtf(vi + o.eps) * gi;
                v[i] = vi;
                m[i] = 0.95f * m[i] + 0.05f * u * u;
                p[i] -= o.lr * u;
            } break;
            case PSEG_OPT_ADAMAX: {
                const float mi = o.b1 * m[i] + (1.0f - o.b1) * gi;
                const float vi = fmaxf(o.b2 * v[i], fabsf(gi));
                m[i] = mi; v[i] = vi;
                p[i] -= o.lr * mi / (vi + o.eps);             // o.lr = lr / (1 - b1^t)
            } break;
            case PSEG_OPT_NADAM: {
                const float gp = gi * o.inv_ms_new;
                const float mi = o.b1 * m[i] + (1.0f - o.b1) * gi;
                const float vi = o.b2 * v[i] + (1.0f - o.b2) * gi * gi;
                m[i] = mi; v[i] = vi;
                const float mbar = (1.0f - o.u_t) * gp + o.u_t1 * (mi * o.inv_ms_next);
                p[i] -= o.lr * mbar / (sqrtf(vi * o.inv_b2t) + o.eps);
            } break;
        }
    }
}

__global__ void fill_kernel(float* p, int64_t n, float v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

// ---------------------------------------------------------------------------------------------
// host orchestration
// ---------------------------------------------------------------------------------------------
static int ensure_buf(void** p, size_t* cap, size_t bytes) {
    if (*cap >= bytes && *p) return PSEG_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    PSEG_HIP(hipMalloc(p, bytes));
    *cap = bytes;
    return PSEG_OK;
}

static int producer_of(const Engine& e, int tensor) {
    for (size_t i = 0; i < e.ops.size(); ++i)
        if (e.ops[i].dst == tensor) return (int)i;
    return -1;
}

static int train_init(Engine& e, float b1, float b2, float eps, float clipnorm, float clipvalue) {
    if (e.mode != PSEG_MODE_F32_EXACT) return fail(PSEG_EUNSUPPORTED, "training runs on the float32 engine (mode F32_EXACT)");
    if (e.n_classes > PSEG_MAXC) return fail(PSEG_EUNSUPPORTED, "training supports at most %d classes", PSEG_MAXC);
    train_free(e);
    auto* t = new TrainState();
    e.train = t;
    t->beta1 = b1; t->beta2 = b2; t->eps = eps; t->clipnorm = clipnorm; t->clipvalue = clipvalue;
    int64_t o = 0;
    for (auto& p : e.params) {
        t->off.push_back(o);
        o += (int64_t)p.host.size();
        o = (o + 3) & ~(int64_t)3;
    }
    t->nparam = o;
    t->nflat = o + 2 + 2 * PSEG_MAXC + 1;   // + the hinge / focal loss sum
    PSEG_HIP(hipMalloc((void**)&t->d_grad, (size_t)t->nflat * 4));
    PSEG_HIP(hipMalloc((void**)&t->d_m, (size_t)t->nparam * 4));
    PSEG_HIP(hipMalloc((void**)&t->d_v, (size_t)t->nparam * 4));
    PSEG_HIP(hipMalloc((void**)&t->d_norm, e.params.size() * 4));
    PSEG_HIP(hipMemset(t->d_m, 0, (size_t)t->nparam * 4));
    PSEG_HIP(hipMemset(t->d_v, 0, (size_t)t->nparam * 4));
    PSEG_HIP(hipDeviceSynchronize());   // null-stream memsets are not ordered with the engine's non-blocking stream
    t->tgrad.assign(e.tensors.size(), nullptr);
    t->tbytes.assign(e.tensors.size(), 0);
    return PSEG_OK;
}

// forward + loss/metrics (+ backward when `backward`); inputs are host pointers
static int train_fwd_bwd(Engine& e, const uint8_t* img, const uint8_t* mask, int H, int W, bool backward,
                         const float* img_f32 = nullptr) {
    TrainState* t = TS(e);
    if (!t) return fail(PSEG_EINVAL, "pseg_train_init has not been called");
    PSEG_HIP(hipSetDevice(e.device));
    for (auto& p : e.params)
        if (!p.set) return fail(PSEG_EINVAL, "weight '%s' was never set", p.name.c_str());
    if (e.weights_dirty) PSEG_TRY(upload_weights(e));
    PSEG_TRY(set_canvas(e, H, W, e.stream));
    hipStream_t st = e.stream;
    const int C = e.n_classes;
    const size_t npx = (size_t)H * W;
    PSEG_TRY(ensure_buf((void**)&t->d_img, &t->img_bytes, npx * e.in_ch * (img_f32 ? 4 : 1)));
    PSEG_TRY(ensure_buf((void**)&t->d_mask, &t->mask_bytes, npx));
    size_t lb = t->logits_bytes;
    PSEG_TRY(ensure_buf((void**)&t->d_logits, &lb, npx * C * 4));
    lb = t->logits_bytes;
    PSEG_TRY(ensure_buf((void**)&t->d_dlogits, &lb, npx * C * 4));
    t->logits_bytes = lb;
    t->H = H; t->W = W;
    if (img_f32) PSEG_HIP(hipMemcpyAsync(t->d_img, img_f32, npx * e.in_ch * 4, hipMemcpyHostToDevice, st));
    else PSEG_HIP(hipMemcpyAsync(t->d_img, img, npx * e.in_ch, hipMemcpyHostToDevice, st));
    PSEG_HIP(hipMemcpyAsync(t->d_mask, mask, npx, hipMemcpyHostToDevice, st));
    e.cur_img_f32 = img_f32 ? (const float*)t->d_img : nullptr;
    // Dropout layers are live in a training forward (Keras fit), the identity in an evaluation step
    const uint32_t drop_key = backward ? (t->drop_seed * 0x632BE5ABu + (uint32_t)t->fwd_count * 0x9E3779B9u) | 1u : 0u;
    if (backward) ++t->fwd_count;
    e.drop_key = drop_key;
    e.bn_training = backward;   // Keras fit: BatchNormalization layers normalise with the batch statistics; evaluate / predict with the moving ones
    e.relaxed_f32 = PSEG_KNOB("PSEG_TRAIN_STRICT") ? 0 : 1;   // wide layers: channel-blocked matrix-core kernel (summation order differs from predict)
    const int rc_fwd = run_exact(e, t->d_img, t->d_logits, nullptr, nullptr, nullptr, st);
    e.drop_key = 0;
    e.bn_training = false;
    e.relaxed_f32 = 0;
    e.cur_img_f32 = nullptr;
    PSEG_TRY(rc_fwd);
    float* acc = t->d_grad + t->nparam;
    // a backward pass starts from zeroed parameter gradients; an evaluation step only resets the metric slots
    if (backward) PSEG_HIP(hipMemsetAsync(t->d_grad, 0, (size_t)t->nflat * 4, st));
    else PSEG_HIP(hipMemsetAsync(acc, 0, (size_t)(t->nflat - t->nparam) * 4, st));
    {
        const int nblk = cdiv((int)npx, 256), nslot = 2 + 2 * C;
        PSEG_TRY(ensure_buf((void**)&t->d_mpart, &t->mpart_bytes, (size_t)nblk * nslot * 4));
        ce_metrics_kernel<<<nblk, 256, 0, st>>>(t->d_logits, t->d_mask, (int)npx, C, 1.0f / (float)npx, t->d_dlogits, t->d_mpart);
        metrics_final_kernel<<<nslot, 256, 0, st>>>(t->d_mpart, nblk, nslot, acc);
    }
    if (t->loss_kind != PSEG_LOSS_CE)
        loss_grad_kernel<<<cdiv((int)npx, 256), 256, 0, st>>>(t->loss_kind, t->d_logits, t->d_mask, (int)npx, C, 1.0f / (float)npx,
                                                            acc, t->d_dlogits, acc + 2 + 2 * PSEG_MAXC);
    PSEG_HIP(hipGetLastError());
    if (!backward) return PSEG_OK;

    // tensor gradients (canvas dims).  They are not zeroed: the first consumer met on the way back STORES its contribution
    // (fresh[i]), the later ones accumulate -- a memset per tensor and a read of the zeros by the first writer were 0.3 ms of a
    // 15 ms step.  Writers without a store form (and a gradient nobody wrote before it is read) zero the buffer first.
    std::vector<char> fresh(e.tensors.size(), 1);
    auto tbytes_of = [&](int i) { const Tensor& tn = e.tensors[i]; return (size_t)e.tH(tn) * e.tW(tn) * tn.C * 4; };
    auto zero_if_fresh = [&](int i) -> int {
        if (i >= 0 && fresh[i]) { PSEG_HIP(hipMemsetAsync(t->tgrad[i], 0, tbytes_of(i), st)); fresh[i] = 0; }
        return PSEG_OK;
    };
    for (size_t i = 0; i < e.tensors.size(); ++i) {
        if ((int)i == e.input_tensor) continue;
        const Tensor& tn = e.tensors[i];
        const size_t bytes = (size_t)e.tH(tn) * e.tW(tn) * tn.C * 4;
        if (bytes > t->tbytes[i]) {
            (void)hipFree(t->tgrad[i]);
            t->tgrad[i] = nullptr;
            PSEG_HIP(hipMalloc((void**)&t->tgrad[i], bytes));
            t->tbytes[i] = bytes;
            // a new buffer starts as NaNs (once, not per step): a region no writer covers then shows in every gradient
            // test instead of depending on what the allocator hands back
            PSEG_HIP(hipMemsetAsync(t->tgrad[i], 0xFF, bytes, st));
        }
    }
    auto ensure_wd = [&](size_t floats) -> int {
        return ensure_buf((void**)&t->d_wd, &t->wd_bytes, (floats + COT) * 4);
    };
    const int strips_target = 1536;
    const bool scalar_wgrad = false;
    // Second stream for the weight gradients.  A layer's weight gradient and its data gradient both only READ the layer's output
    // gradient, and nothing downstream of a weight gradient runs before the optimizer: the weight-gradient kernels (and their
    // ordered reductions) go to `ws`, gated by an event recorded on the main stream once dY is final; the main stream carries on
    // with the data gradients and joins at the end.  Two MFMA-bound kernels of 65-85 % pipe-busy each fill each other's gaps.
    hipStream_t ws = st;
    if (!PSEG_KNOB("PSEG_TRAIN_ONE_STREAM")) {
        if (!t->wstream) {
            PSEG_HIP(hipStreamCreateWithFlags(&t->wstream, hipStreamNonBlocking));
            PSEG_HIP(hipEventCreateWithFlags(&t->ev_dy, hipEventDisableTiming));
            PSEG_HIP(hipEventCreateWithFlags(&t->ev_wdone, hipEventDisableTiming));
        }
        ws = t->wstream;
    }
    // every way out of this function -- also an early error return -- leaves the main stream waiting for what was enqueued on `ws`:
    // the next step's zeroing of the gradient buffer and its forward must not overtake weight-gradient kernels still in flight
    struct Join {
        hipStream_t ws, st; hipEvent_t ev; bool done = false;
        void now() { if (!done && ws != st && ev) { (void)hipEventRecord(ev, ws); (void)hipStreamWaitEvent(st, ev, 0); } done = true; }
        ~Join() { now(); }
    } join{ws, st, t->ev_wdone};
    auto dy_ready = [&]() -> int {      // everything enqueued on the main stream so far is visible to the next kernel on ws
        if (ws != st) { PSEG_HIP(hipEventRecord(t->ev_dy, st)); PSEG_HIP(hipStreamWaitEvent(ws, t->ev_dy, 0)); }
        return PSEG_OK;
    };

    for (int oi = (int)e.ops.size() - 1; oi >= 0; --oi) {
        Op& op = e.ops[oi];
        const Tensor& s0 = e.tensors[op.src0];
        const Tensor* s1 = op.src1 >= 0 ? &e.tensors[op.src1] : nullptr;
        const int C0 = s0.C, C1 = s1 ? s1->C : 0;
        float* gw = op.kparam >= 0 ? t->d_grad + t->off[op.kparam] : nullptr;
        float* gb = op.bparam >= 0 ? t->d_grad + t->off[op.bparam] : nullptr;
        if (op.type != OP_LOGITS) PSEG_TRY(zero_if_fresh(op.dst));   // (a tensor nothing consumed: its gradient is zero)
        if (op.dropout > 0.0f && drop_key) {   // same mask and scale as the forward, on the gradient of the dropped tensor
            const Tensor& d = e.tensors[op.dst];
            launch_dropout(t->tgrad[op.dst], (size_t)e.tH(d) * e.tW(d) * d.C, drop_key + 0x85EBCA77u * (uint32_t)oi, op.dropout, st);
        }
        if (op.type == OP_LOGITS || op.type == OP_CONV) {
            if (op.stride != 1 && op.stride != 2) return fail(PSEG_EUNSUPPORTED, "backward of layer %s (stride %d) is not built", op.layer.c_str(), op.stride);
            const bool lg = op.type == OP_LOGITS;
            const float* dY = lg ? t->d_dlogits : t->tgrad[op.dst];
            const float* Y = lg ? nullptr : (const float*)e.tensors[op.dst].d;
            const float* maskY = (op.relu && Y) ? Y : nullptr;
            const int Hy = lg ? H : e.tH(e.tensors[op.dst]), Wy = lg ? W : e.tW(e.tensors[op.dst]);
            // extent of the conv's input: both concat sources share it (a half-resolution source is read through up0 / up1)
            const int Hx = e.tH(s0) << op.up0, Wx = e.tW(s0) << op.up0;
            const int k = op.k, st_ = op.stride;
            int pt = 0, pl = 0;
            if (!lg) {   // TF SAME: pad_total = max((out-1)*s + k - in, 0), before = total / 2 (as run_exact)
                pt = std::max((Hy - 1) * st_ + k - Hx, 0) / 2;
                pl = std::max((Wy - 1) * st_ + k - Wx, 0) / 2;
            }
            // residual addend (Add() after the bias): its gradient is the layer's output gradient
            if (op.add >= 0) {
                const Tensor& ad = e.tensors[op.add];
                const size_t n = (size_t)e.tH(ad) * e.tW(ad) * ad.C;
                PSEG_TRY(zero_if_fresh(op.add));
                accum_kernel<<<(int)std::min<size_t>((n + 255) / 256, 8192), 256, 0, st>>>(t->tgrad[op.add], dY, maskY, nullptr, n);
            }
            // ---- wgrad + bias grad ----
            bool paired = false;
            if (lg && !scalar_wgrad) {   // the 1x1 logits layer: both sources and the bias in one pass over dY
                WgradArgs a{};
                a.X = (const float*)s0.d; a.XC0 = C0; a.X1 = s1 ? (const float*)s1->d : nullptr; a.XC = C0 + C1; a.ci0 = 0;
                a.Hx = Hx; a.Wx = Wx; a.xpitch = Wx; a.xup = 0; a.stride = 1; a.in_relu = op.in_relu;
                a.dY = dY; a.maskY = maskY; a.Hy = Hy; a.Wy = Wy; a.ypitch = Wy; a.Cout = op.Cout; a.Cin = op.Cin;
                a.KW = k; a.pt = pt; a.pl = pl; a.mode = 0; a.strip_rows = 1; a.dW = gw; a.dB = gb;
                WgradFlatPlan pp;
                if (!op.up0 && !op.up1 && st_ == 1 && wgrad_pair_plan(a, k * k, &pp)) {
                    PSEG_TRY(dy_ready());
                    PSEG_TRY(launch_wgrad(a, dim3(1, k * k), ws, &t->d_wpart, &t->wpart_bytes));
                    paired = true;
                }
            }
            for (int sidx = 0; sidx < (paired ? 0 : (s1 ? 2 : 1)); ++sidx) {
                const Tensor& sx = sidx == 0 ? s0 : *s1;
                const int up = sidx == 0 ? op.up0 : op.up1;
                WgradArgs a{};
                a.X = (const float*)sx.d; a.XC = sx.C; a.ci0 = sidx == 0 ? 0 : C0; a.Hx = Hx; a.Wx = Wx; a.xpitch = Wx >> up;
                a.xup = up; a.stride = st_; a.in_relu = op.in_relu;
                a.dY = dY; a.maskY = maskY;
                a.Hy = Hy; a.Wy = Wy; a.ypitch = Wy; a.Cout = op.Cout; a.Cin = op.Cin;
                a.KW = k; a.pt = pt; a.pl = pl; a.mode = 0;
                a.strip_rows = std::max(1, cdiv(Hy * k * k, strips_target));
                a.dW = gw; a.dB = sidx == 0 ? gb : nullptr;
                dim3 grid(cdiv(Hy, a.strip_rows), k * k);
                PSEG_TRY(dy_ready());
                PSEG_TRY(launch_wgrad(a, grid, ws, &t->d_wpart, &t->wpart_bytes));
            }
            // ---- dgrad into the source gradients (skipped for the network input) ----
            // = a stride-1 convolution of the (ReLU-masked, for stride 2 zero-dilated) output gradient with the flipped
            // kernel at the conv's input extent.  Plain sources accumulate in place; an upsampled source gets the sum of
            // its 2x2 blocks, a pre-activation source the X > 0 mask -- both through a scratch tensor.
            const float* dYd = dY;
            const float* maskd = maskY;
            if (st_ == 2) {
                const size_t bytes = (size_t)Hx * Wx * op.Cout * 4;
                PSEG_TRY(ensure_buf((void**)&t->d_tmp2, &t->tmp2_bytes, bytes));
                const size_t n = bytes / 4;
                dilate2_kernel<<<(int)std::min<size_t>((n + 255) / 256, 8192), 256, 0, st>>>(t->d_tmp2, dY, maskY, Hy, Wy, op.Cout);
                dYd = t->d_tmp2;
                maskd = nullptr;
            }
            for (int sidx = 0; sidx < (s1 ? 2 : 1); ++sidx) {
                const int src = sidx == 0 ? op.src0 : op.src1;
                if (src == e.input_tensor) continue;
                const int up = sidx == 0 ? op.up0 : op.up1;
                const int c0 = sidx == 0 ? 0 : C0, nc = sidx == 0 ? C0 : C1;
                const bool direct = !up && !op.in_relu;
                PSEG_TRY(ensure_wd((size_t)k * k * op.Cout * nc));
                wd_conv_kernel<<<(int)std::min<size_t>(((size_t)k * k * op.Cout * nc + 255) / 256, 2048), 256, 0, st>>>(op.d_w, k, k, op.Cin, op.Cout, c0, nc, t->d_wd);   // (64 blocks took 49 us for unet's 4.7 M-element kernels)
                if (!direct) PSEG_TRY(ensure_buf((void**)&t->d_tmp, &t->tmp_bytes, (size_t)Hx * Wx * nc * 4));
                ConvArgs a{};
                a.src0 = dYd; a.C0 = op.Cout; a.Hin = st_ == 2 ? Hx : Hy; a.Win = st_ == 2 ? Wx : Wy;
                a.w = t->d_wd; a.bias = nullptr; a.KH = a.KW = k; a.stride = 1;
                if (const size_t wrb = wrem_bytes_for(k, k, op.Cout, nc)) {   // (the scratch kernel changes with every layer: no validity flag)
                    PSEG_TRY(ensure_buf((void**)&t->d_wdrem, &t->wdrem_bytes, wrb));
                    a.wrem_buf = t->d_wdrem; a.wrem_cap = t->wdrem_bytes;
                }
                a.pt = k - 1 - pt; a.pl = k - 1 - pl;
                a.Hout = lg ? H : Hx; a.Wout = lg ? W : Wx; a.Cout = nc;
                a.mask = maskd;
                a.relaxed = PSEG_KNOB("PSEG_TRAIN_STRICT") ? 0 : 1;
                a.dst = direct ? t->tgrad[src] : t->d_tmp;
                // (the logits layer writes the page extent only: on a padded canvas the rest of the gradient must be zeros)
                if (direct && lg && (H != Hx || W != Wx)) PSEG_TRY(zero_if_fresh(src));
                a.add = (direct && !fresh[src]) ? t->tgrad[src] : nullptr;
                if (direct) fresh[src] = 0; else PSEG_TRY(zero_if_fresh(src));
                a.dst_pitch = Wx;
                PSEG_TRY(launch_conv_exact(a, st));
                if (!direct) {
                    const Tensor& sx = e.tensors[src];
                    const float* maskX = op.in_relu ? (const float*)sx.d : nullptr;
                    const size_t n = (size_t)e.tH(sx) * e.tW(sx) * nc;
                    const int g = (int)std::min<size_t>((n + 255) / 256, 8192);
                    if (up) upsample_bwd_kernel<<<g, 256, 0, st>>>(t->tgrad[src], t->d_tmp, maskX, e.tH(sx), e.tW(sx), nc);
                    else accum_kernel<<<g, 256, 0, st>>>(t->tgrad[src], t->d_tmp, nullptr, maskX, n);
                }
                PSEG_HIP(hipGetLastError());
            }
        } else if (op.type == OP_DECONV2) {
            const float* dY = t->tgrad[op.dst];
            const float* Y = (const float*)e.tensors[op.dst].d;
            const int Hx = e.tH(s0), Wx = e.tW(s0);
            bool paired = false;
            if (!scalar_wgrad) {   // both sources and the bias gradient in one pass over dY
                WgradArgs a{};
                a.X = (const float*)s0.d; a.XC0 = C0; a.X1 = s1 ? (const float*)s1->d : nullptr; a.XC = C0 + C1; a.ci0 = 0;
                a.Hx = Hx; a.Wx = Wx; a.xpitch = Wx;
                a.dY = dY; a.maskY = op.relu ? Y : nullptr;
                a.Hy = 2 * Hx; a.Wy = 2 * Wx; a.ypitch = 2 * Wx; a.Cout = op.Cout; a.Cin = op.Cin;
                a.KW = 2; a.mode = 1; a.strip_rows = 1; a.dW = gw; a.dB = gb;
                WgradFlatPlan pp;
                if (wgrad_pair_plan(a, 4, &pp)) {
                    PSEG_TRY(dy_ready());
                    PSEG_TRY(launch_wgrad(a, dim3(1, 4), ws, &t->d_wpart, &t->wpart_bytes));
                    paired = true;
                }
            }
            for (int sidx = 0; sidx < (paired ? 0 : (s1 ? 2 : 1)); ++sidx) {
                const Tensor& sx = sidx == 0 ? s0 : *s1;
                WgradArgs a{};
                a.X = (const float*)sx.d; a.XC = sx.C; a.ci0 = sidx == 0 ? 0 : C0; a.Hx = Hx; a.Wx = Wx; a.xpitch = Wx;
                a.dY = dY; a.maskY = op.relu ? Y : nullptr;
                a.Hy = 2 * Hx; a.Wy = 2 * Wx; a.ypitch = 2 * Wx; a.Cout = op.Cout; a.Cin = op.Cin;
                a.KW = 2; a.mode = 1;
                a.strip_rows = std::max(1, cdiv(Hx * 4, strips_target));
                a.dW = gw; a.dB = nullptr;
                dim3 grid(cdiv(Hx, a.strip_rows), 4);
                PSEG_TRY(dy_ready());
                PSEG_TRY(launch_wgrad(a, grid, ws, &t->d_wpart, &t->wpart_bytes));
            }
            if (!paired) {
            if (op.Cout > 256) return fail(PSEG_EUNSUPPORTED, "bias gradient supports at most 256 channels");
            PSEG_TRY(ensure_buf((void**)&t->d_bpart, &t->bpart_bytes, (size_t)BG_BLOCKS * op.Cout * 8));
            PSEG_TRY(dy_ready());
            bias_grad_kernel<<<BG_BLOCKS, 256, 0, ws>>>(dY, op.relu ? Y : nullptr, (size_t)4 * Hx * Wx, op.Cout, (double*)t->d_bpart);
            column_sum_kernel<double><<<op.Cout, 256, 0, ws>>>((const double*)t->d_bpart, BG_BLOCKS, op.Cout, gb);
            }
            PSEG_HIP(hipGetLastError());
            for (int sidx = 0; sidx < (s1 ? 2 : 1); ++sidx) {
                const int src = sidx == 0 ? op.src0 : op.src1;
                const int c0 = sidx == 0 ? 0 : C0, nc = sidx == 0 ? C0 : C1;
                PSEG_TRY(ensure_wd((size_t)4 * op.Cout * nc));
                wd_deconv_kernel<<<(int)std::min<size_t>(((size_t)4 * op.Cout * nc + 255) / 256, 2048), 256, 0, st>>>(op.d_w, op.Cin, op.Cout, c0, nc, t->d_wd);
                // = a k2 stride-2 convolution of the (ReLU-masked) output gradient with wd[ab][co][c],
                // accumulated in place: the matrix-core kernel when its tile fits, else the scalar one
                ConvArgs c{};
                c.src0 = dY; c.C0 = op.Cout; c.mask = op.relu ? Y : nullptr;
                c.Hin = 2 * Hx; c.Win = 2 * Wx; c.Hout = Hx; c.Wout = Wx;
                c.w = t->d_wd; c.KH = c.KW = 2; c.stride = 2; c.Cout = nc;
                c.add = fresh[src] ? nullptr : t->tgrad[src]; c.dst = t->tgrad[src];
                c.relaxed = 1;
                const int rc = launch_conv_exact_mfma(c, st);
                if (rc < 0) return rc;
                if (rc != 0) fresh[src] = 0;
                if (rc == 0) {
                    PSEG_TRY(zero_if_fresh(src));
                    dim3 grid(cdiv(Hx * Wx, 256), cdiv(nc, COT));
                    deconv2_dgrad_kernel<<<grid, 256, 0, st>>>(dY, Y, op.relu, Hx, Wx, op.Cout, t->d_wd, nc, t->tgrad[src]);
                }
                PSEG_HIP(hipGetLastError());
            }
        } else if (op.type == OP_BN) {
            // dgamma / dbeta into this op's channel slice of the layer's gradient vectors, dx accumulated into the source
            const float* Y = (const float*)e.tensors[op.dst].d;
            if (op.src0 != e.input_tensor) PSEG_TRY(zero_if_fresh(op.src0));
            PSEG_TRY(bn_backward((const float*)s0.d, op.relu ? Y : nullptr, t->tgrad[op.dst],
                                 op.src0 == e.input_tensor ? nullptr : t->tgrad[op.src0], (size_t)e.tH(s0) * e.tW(s0), op.Cin, op.d_w,
                                 op.d_b, gw + op.bn_c0, gb + op.bn_c0, st));
        } else if (op.type == OP_POOL) {
            const size_t n = (size_t)(e.tH(s0) / 2) * (e.tW(s0) / 2) * s0.C;
            pool_bwd_kernel<<<(int)std::min<size_t>((n + 255) / 256, 8192), 256, 0, st>>>(
                (const float*)s0.d, t->tgrad[op.dst], e.tH(s0), e.tW(s0), s0.C, t->tgrad[op.src0], (int)fresh[op.src0]);
            fresh[op.src0] = 0;
            PSEG_HIP(hipGetLastError());
        }
    }
    (void)producer_of;
    join.now();   // the optimizer / all-reduce / readers see every gradient
    return PSEG_OK;
}

static int train_metrics(Engine& e, float out[4]) {
    TrainState* t = TS(e);
    const int C = e.n_classes;
    std::vector<float> acc(2 + 2 * PSEG_MAXC + 1);
    PSEG_HIP(hipStreamSynchronize(e.stream));
    PSEG_HIP(hipMemcpy(acc.data(), t->d_grad + t->nparam, acc.size() * 4, hipMemcpyDeviceToHost));
    const double n = (double)t->H * t->W;
    out[0] = (float)(acc[0] / n);
    out[1] = (float)(acc[1] / n);
    double jac = 0, dice = 0;
    for (int c = 0; c < C; ++c) {
        const double I = acc[2 + c], S = acc[2 + C + c];
        jac += (I + 100.0) / (S - I + 100.0);
        dice += (2.0 * I + 100.0) / (S + 100.0);
    }
    out[2] = (float)(jac / C);
    out[3] = (float)(dice / C);
    // metrics[0] is the compiled loss (Keras reports `loss`)
    if (t->loss_kind == PSEG_LOSS_DICE || t->loss_kind == PSEG_LOSS_JACCARD || t->loss_kind == PSEG_LOSS_DICE_CE) {
        double l = 0;
        for (int c = 0; c < C; ++c) {
            const double I = acc[2 + c], S = acc[2 + C + c];
            l += -std::log(t->loss_kind == PSEG_LOSS_JACCARD ? (I + 100.0) / (S - I + 100.0) : (2.0 * I + 100.0) / (S + 100.0));
        }
        out[0] = (float)(l / C * (t->loss_kind == PSEG_LOSS_DICE_CE ? 0.5 : 1.0));
    } else if (t->loss_kind == PSEG_LOSS_HINGE) {
        out[0] = (float)(acc[2 + 2 * PSEG_MAXC] / n);
    } else if (t->loss_kind == PSEG_LOSS_FOCAL) {
        out[0] = (float)(acc[2 + 2 * PSEG_MAXC] / n);
    }
    return PSEG_OK;
}

static int train_apply(Engine& e, float lr, float gscale) {
    TrainState* t = TS(e);
    if (!t) return fail(PSEG_EINVAL, "pseg_train_init has not been called");
    PSEG_HIP(hipSetDevice(e.device));
    hipStream_t st = e.stream;
    t->step += 1;
    OptScalars o_{};
    OptScalars& o = o_;
    o.lr = lr; o.b1 = t->beta1; o.b2 = t->beta2; o.eps = t->eps;
    const double b1t = std::pow((double)t->beta1, (double)t->step), b2t = std::pow((double)t->beta2, (double)t->step);
    if (t->optimizer == PSEG_OPT_ADAM) o.lr = (float)(lr * std::sqrt(1.0 - b2t) / (1.0 - b1t));
    if (t->optimizer == PSEG_OPT_ADAMAX) o.lr = (float)(lr / (1.0 - b1t));
    if (t->optimizer == PSEG_OPT_NADAM) {
        // Keras nadam.py: u_t = b1 (1 - 0.5 * 0.96^(0.004 t)); m_schedule is the running product of the u's
        const double ut = t->beta1 * (1.0 - 0.5 * std::pow(0.96, 0.004 * (double)t->step));
        const double ut1 = t->beta1 * (1.0 - 0.5 * std::pow(0.96, 0.004 * (double)(t->step + 1)));
        const double ms_new = t->m_schedule * ut, ms_next = ms_new * ut1;
        t->m_schedule = ms_new;
        o.inv_ms_new = (float)(1.0 / (1.0 - ms_new));
        o.inv_ms_next = (float)(1.0 / (1.0 - ms_next));
        o.inv_b2t = (float)(1.0 / (1.0 - b2t));
        o.u_t = (float)ut;
        o.u_t1 = (float)ut1;
    }
    if (t->optimizer == PSEG_OPT_ADAGRAD && !t->state_init) {
        fill_kernel<<<1024, 256, 0, st>>>(t->d_v, t->nparam, 0.1f);      // initial_accumulator_value
        t->state_init = true;
    }
    PSEG_HIP(hipMemsetAsync(t->d_norm, 0, e.params.size() * 4, st));
    // parameter -> device buffer (kernels: op.d_w in correlation layout; biases: op.d_b; BatchNormalization: gamma and
    // beta slices at the head of op.d_w -- the moving statistics are not trained).  All norms first: the slices of one
    // BatchNormalization tensor spread over two ops share a clip_by_norm.
    if (!t->d_part) {   // slices in op order: (op, kernel | bias) -- fixed for the life of the engine
        std::vector<int> nblk, pidx;
        for (auto& op : e.ops) {
            if (op.kparam < 0) continue;
            for (int which = 0; which < 2; ++which) {
                const int pi = which == 0 ? op.kparam : op.bparam;
                const int64_t n = op.type == OP_BN ? op.Cin : (int64_t)e.params[pi].host.size();
                nblk.push_back((int)std::min<int64_t>((n + 255) / 256, SUMSQ_MAXB));
                pidx.push_back(pi);
            }
        }
        t->nslices = (int)nblk.size();
        PSEG_HIP(hipMalloc((void**)&t->d_part, (size_t)t->nslices * SUMSQ_MAXB * 4));
        PSEG_HIP(hipMalloc((void**)&t->d_slice_nblk, nblk.size() * 4));
        PSEG_HIP(hipMalloc((void**)&t->d_slice_pi, pidx.size() * 4));
        PSEG_HIP(hipMemcpy(t->d_slice_nblk, nblk.data(), nblk.size() * 4, hipMemcpyHostToDevice));
        PSEG_HIP(hipMemcpy(t->d_slice_pi, pidx.data(), pidx.size() * 4, hipMemcpyHostToDevice));
        PSEG_HIP(hipDeviceSynchronize());
    }
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 0 && !(t->clipnorm > 0.0f)) continue;
        int slice = 0;
        for (auto& op : e.ops) {
            if (op.kparam < 0) continue;
            const bool bn = op.type == OP_BN;
            for (int which = 0; which < 2; ++which, ++slice) {
                const int pi = which == 0 ? op.kparam : op.bparam;
                const int64_t n = bn ? op.Cin : (int64_t)e.params[pi].host.size();
                const int64_t o = t->off[pi] + (bn ? op.bn_c0 : 0);
                float* g = t->d_grad + o;
                float* p = bn ? op.d_w + which * op.Cin : (which == 0 ? op.d_w : op.d_b);
                const int grid = (int)std::min<int64_t>((n + 255) / 256, SUMSQ_MAXB);
                if (pass == 0) sumsq_kernel<<<grid, 256, 0, st>>>(g, n, gscale, t->d_part + (size_t)slice * SUMSQ_MAXB);
                else opt_kernel<<<grid, 256, 0, st>>>(t->optimizer, p, g, t->d_m + o, t->d_v + o, n, gscale,
                                                     t->d_norm + pi, t->clipnorm, t->clipvalue, o_);
            }
        }
        if (pass == 0) sumsq_final_kernel<<<t->nslices, 256, 0, st>>>(t->d_part, t->d_slice_nblk, t->d_slice_pi, t->d_norm);
    }
    for (auto& op : e.ops) op.wrem_valid = false;   // the kernels changed: the float32 convs rebuild their left-over channel copies
    PSEG_HIP(hipGetLastError());
    return PSEG_OK;
}

// device (correlation) layout -> Keras layout, inverse of upload_weights()
static void to_keras(const Op& op, const std::vector<float>& w, std::vector<float>& out) {
    const int k = op.k, Cin = op.Cin, Cout = op.Cout;
    for (int ky = 0; ky < k; ++ky)
        for (int kx = 0; kx < k; ++kx)
            for (int ci = 0; ci < Cin; ++ci)
                for (int co = 0; co < Cout; ++co) {
                    const float v = w[(((size_t)ky * k + kx) * Cin + ci) * Cout + co];
                    size_t o;
                    if (!op.transposed) o = (((size_t)ky * k + kx) * Cin + ci) * Cout + co;
                    else if (op.type == OP_DECONV2) o = (((size_t)ky * k + kx) * Cout + co) * Cin + ci;
                    else o = (((size_t)(k - 1 - ky) * k + (k - 1 - kx)) * Cout + co) * Cin + ci;
                    out[o] = v;
                }
}

int train_sync_weights_to_host(Engine& e) {
    PSEG_HIP(hipSetDevice(e.device));
    PSEG_HIP(hipStreamSynchronize(e.stream));
    for (auto& op : e.ops) {
        if (op.kparam < 0 || !op.d_w) continue;
        if (op.type == OP_BN) {
            const int pidx[4] = {op.kparam, op.bparam, op.mmparam, op.mvparam};
            for (int j = 0; j < 4; ++j)
                PSEG_HIP(hipMemcpy(e.params[pidx[j]].host.data() + op.bn_c0, op.d_w + (size_t)j * op.Cin, (size_t)op.Cin * 4, hipMemcpyDeviceToHost));
            continue;
        }
        Param& kp = e.params[op.kparam];
        Param& bp = e.params[op.bparam];
        std::vector<float> w(kp.host.size());
        PSEG_HIP(hipMemcpy(w.data(), op.d_w, w.size() * 4, hipMemcpyDeviceToHost));
        to_keras(op, w, kp.host);
        PSEG_HIP(hipMemcpy(bp.host.data(), op.d_b, bp.host.size() * 4, hipMemcpyDeviceToHost));
    }
    return PSEG_OK;
}

}  // namespace pseg

using namespace pseg;

extern "C" {

int pseg_train_init(pseg_engine* h, float beta1, float beta2, float eps, float clipnorm, float clipvalue) {
    if (!h) return fail(PSEG_EINVAL, "NULL engine");
    KnobScope knob_scope(h->e);
    PSEG_HIP(hipSetDevice(h->e.device));
    return train_init(h->e, beta1, beta2, eps, clipnorm, clipvalue);
}

int pseg_train_set_optimizer(pseg_engine* h, int optimizer) {
    if (!h) return fail(PSEG_EINVAL, "NULL engine");
    if (optimizer < PSEG_OPT_ADAM || optimizer > PSEG_OPT_NADAM) return fail(PSEG_EINVAL, "unknown optimizer id %d", optimizer);
    TrainState* t = TS(h->e);
    if (!t) return fail(PSEG_EINVAL, "pseg_train_init has not been called");
    PSEG_HIP(hipSetDevice(h->e.device));
    PSEG_HIP(hipStreamSynchronize(h->e.stream));
    PSEG_HIP(hipMemset(t->d_m, 0, (size_t)t->nparam * 4));
    PSEG_HIP(hipMemset(t->d_v, 0, (size_t)t->nparam * 4));
    PSEG_HIP(hipDeviceSynchronize());   // null-stream memsets are not ordered with the engine's non-blocking stream
    t->optimizer = optimizer;
    t->step = 0;
    t->m_schedule = 1.0;
    t->state_init = false;
    return PSEG_OK;
}

int pseg_train_set_dropout_seed(pseg_engine* h, uint32_t seed) {
    if (!h || !h->e.train) return fail(PSEG_EINVAL, "pseg_train_init has not been called");
    TrainState* t = (TrainState*)h->e.train;
    t->drop_seed = seed;
    t->fwd_count = 0;
    return PSEG_OK;
}

int pseg_train_set_loss(pseg_engine* h, int loss) {
    if (!h) return fail(PSEG_EINVAL, "NULL engine");
    if (loss < PSEG_LOSS_CE || loss > PSEG_LOSS_DICE_CE) return fail(PSEG_EINVAL, "unknown loss id %d", loss);
    TrainState* t = TS(h->e);
    if (!t) return fail(PSEG_EINVAL, "pseg_train_init has not been called");
    if (h->e.n_classes > PSEG_MAXC) return fail(PSEG_EUNSUPPORTED, "losses other than cross-entropy support at most %d classes", PSEG_MAXC);
    t->loss_kind = loss;
    return PSEG_OK;
}

int pseg_train_forward_backward(pseg_engine* h, const uint8_t* img, const uint8_t* mask, int H, int W, float metrics[4]) {
    if (!h || !img || !mask) return fail(PSEG_EINVAL, "NULL argument");
    KnobScope knob_scope(h->e);
    if (H <= 0 || W <= 0) return fail(PSEG_EINVAL, "empty page %dx%d", H, W);
    PSEG_TRY(train_fwd_bwd(h->e, img, mask, H, W, true));
    if (metrics) PSEG_TRY(train_metrics(h->e, metrics));
    return PSEG_OK;
}

int pseg_train_forward_backward_f32(pseg_engine* h, const float* img, const uint8_t* mask, int H, int W, float metrics[4]) {
    if (!h || !img || !mask) return fail(PSEG_EINVAL, "NULL argument");
    KnobScope knob_scope(h->e);
    if (H <= 0 || W <= 0) return fail(PSEG_EINVAL, "empty page %dx%d", H, W);
    PSEG_TRY(train_fwd_bwd(h->e, nullptr, mask, H, W, true, img));
    if (metrics) PSEG_TRY(train_metrics(h->e, metrics));
    return PSEG_OK;
}

int pseg_eval_step(pseg_engine* h, const uint8_t* img, const uint8_t* mask, int H, int W, float metrics[4]) {
    if (!h || !img || !mask || !metrics) return fail(PSEG_EINVAL, "NULL argument");
    KnobScope knob_scope(h->e);
    if (H <= 0 || W <= 0) return fail(PSEG_EINVAL, "empty page %dx%d", H, W);
    if (!h->e.train) PSEG_TRY(train_init(h->e, 0.9f, 0.999f, 1e-7f, 0.0f, 0.0f));
    PSEG_TRY(train_fwd_bwd(h->e, img, mask, H, W, false));
    return train_metrics(h->e, metrics);
}

int pseg_train_grad_buffer(pseg_engine* h, float** d_grad, int64_t* count) {
    if (!h || !h->e.train) return fail(PSEG_EINVAL, "pseg_train_init has not been called");
    TrainState* t = (TrainState*)h->e.train;
    if (d_grad) *d_grad = t->d_grad;
    if (count) *count = t->nflat;
    return PSEG_OK;
}

int pseg_train_metrics(pseg_engine* h, float metrics[4]) {
    if (!h || !h->e.train || !metrics) return fail(PSEG_EINVAL, "bad argument");
    return train_metrics(h->e, metrics);
}

int pseg_train_apply(pseg_engine* h, float lr, float grad_scale) {
    if (!h) return fail(PSEG_EINVAL, "NULL engine");
    KnobScope knob_scope(h->e);
    return train_apply(h->e, lr, grad_scale);
}

int pseg_train_get_gradient(pseg_engine* h, const char* name, float* out, int64_t count) {
    if (!h || !h->e.train || !name || !out) return fail(PSEG_EINVAL, "bad argument");
    Engine& e = h->e;
    TrainState* t = (TrainState*)e.train;
    PSEG_HIP(hipSetDevice(e.device));
    for (size_t pi = 0; pi < e.params.size(); ++pi) {
        if (e.params[pi].name != name) continue;
        const int64_t n = (int64_t)e.params[pi].host.size();
        if (count != n) return fail(PSEG_EINVAL, "gradient '%s' has %lld elements", name, (long long)n);
        PSEG_HIP(hipStreamSynchronize(e.stream));
        std::vector<float> g((size_t)n);
        PSEG_HIP(hipMemcpy(g.data(), t->d_grad + t->off[pi], (size_t)n * 4, hipMemcpyDeviceToHost));
        for (auto& op : e.ops)
            if (op.kparam == (int)pi && op.type != OP_BN) {
                std::vector<float> k((size_t)n);
                to_keras(op, g, k);
                std::copy(k.begin(), k.end(), out);
                return PSEG_OK;
            }
        std::copy(g.begin(), g.end(), out);   // bias
        return PSEG_OK;
    }
    return fail(PSEG_ENOTFOUND, "no weight named '%s'", name);
}

}  // extern "C"
