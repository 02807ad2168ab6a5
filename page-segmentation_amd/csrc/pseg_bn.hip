// pseg_bn.hip -- tf.keras.layers.BatchNormalization of the residual U-Net's bn_act (lib/model.py:265-271) and of
// conv_block_simple (lib/model.py:310-317), NHWC float32 (+ the bf16 inference form).
//
// Inference (moving statistics):  y = (x - mean) * (gamma / sqrt(var + eps)) + beta        -- elementwise, bit-exact
// with the oracle's numpy restatement (correctly rounded sqrt / divide, no contraction).
// Training (batch statistics over N*H*W, N = 1 page on the padded canvas, as the Lambda pad sits in front of every
// layer, lib/model.py:276-277):
//     mu = mean(x), var = mean((x - mu)^2), xhat = (x - mu) / sqrt(var + eps), y = xhat * gamma + beta
//     moving_mean -= (moving_mean - mu) * (1 - momentum);  moving_var likewise with the UNBIASED batch variance
//     (the fused kernel Keras picks for 4-D NHWC inputs applies Bessel's correction to the running value only)
//     dbeta = sum(dy), dgamma = sum(dy * xhat), dx = gamma / sqrt(var + eps) * (dy - dbeta / n - xhat * dgamma / n)
// The per-channel reductions are HBM-bound passes: every thread keeps one channel (consecutive lanes = consecutive
// channels, so a wave reads 256 contiguous bytes), accumulates its pixel stripe in double, the block folds its pixel
// lanes through LDS and issues one double atomic per channel.
#include <algorithm>

#include "pseg_common.h"

namespace pseg {

namespace {

constexpr int BN_T = 256;

struct Lane {
    int c;        // channel of this thread (-1: idle)
    int p, ppb;   // pixel lane, pixel lanes per block
};

__device__ __forceinline__ Lane bn_lane(int C) {
    const int cpb = C < BN_T ? C : BN_T;
    Lane l;
    l.ppb = BN_T / cpb;
    l.p = (int)threadIdx.x / cpb;
    const int c = (int)blockIdx.y * BN_T + (int)threadIdx.x % cpb;
    l.c = (l.p < l.ppb && c < C) ? c : -1;
    return l;
}

// block fold of `v` over the pixel lanes of each channel, then one atomic per channel into out[c]
template <int N>
__device__ __forceinline__ void bn_fold(double (&v)[N], const Lane& l, int C, double* const (&out)[N]) {
    __shared__ double sh[N][BN_T];
#pragma unroll
    for (int i = 0; i < N; ++i) sh[i][threadIdx.x] = l.c >= 0 ? v[i] : 0.0;
    __syncthreads();
    if (l.c >= 0 && l.p == 0) {
        const int cpb = C < BN_T ? C : BN_T;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double s = 0.0;
            for (int q = 0; q < l.ppb; ++q) s += sh[i][q * cpb + (int)threadIdx.x];
            atomicAdd(out[i] + l.c, s);
        }
    }
}

__global__ void __launch_bounds__(BN_T) bn_sum_kernel(const float* __restrict__ x, size_t npx, int C, double* __restrict__ sum) {
    const Lane l = bn_lane(C);
    double v[1] = {0.0};
    if (l.c >= 0)
        for (size_t px = (size_t)blockIdx.x * l.ppb + l.p; px < npx; px += (size_t)gridDim.x * l.ppb) v[0] += (double)x[px * C + l.c];
    double* const out[1] = {sum};
    bn_fold<1>(v, l, C, out);
}

__global__ void __launch_bounds__(BN_T) bn_sqdev_kernel(const float* __restrict__ x, size_t npx, int C, const double* __restrict__ sum,
                                                       double* __restrict__ sq) {
    const Lane l = bn_lane(C);
    double v[1] = {0.0};
    if (l.c >= 0) {
        const double mu = (double)(float)(sum[l.c] / (double)npx);
        for (size_t px = (size_t)blockIdx.x * l.ppb + l.p; px < npx; px += (size_t)gridDim.x * l.ppb) {
            const double d = (double)x[px * C + l.c] - mu;
            v[0] += d * d;
        }
    }
    double* const out[1] = {sq};
    bn_fold<1>(v, l, C, out);
}

// batch statistics -> saved [mean | invstd], moving statistics update
// (`n_seen` = samples the layer sees: npx times 4 per UpSampling2D folded into the consumer -- same mean and variance,
// a different Bessel factor)
__global__ void bn_finalize_kernel(const double* __restrict__ sum, const double* __restrict__ sq, size_t npx, double n_seen, int C, float eps,
                                   float momentum, float* __restrict__ par, float* __restrict__ saved) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double n = (double)npx;
    const float mu = (float)(sum[c] / n);
    const float var = (float)(sq[c] / n);
    saved[c] = mu;
    saved[C + c] = __fdiv_rn(1.0f, __fsqrt_rn(var + eps));
    const float unbiased = n_seen > 1.0 ? (float)(sq[c] / n * (n_seen / (n_seen - 1.0))) : var;
    float* mm = par + 2 * C;
    float* mv = par + 3 * C;
    const float dec = 1.0f - momentum;
    mm[c] = mm[c] - (mm[c] - mu) * dec;
    mv[c] = mv[c] - (mv[c] - unbiased) * dec;
}

__global__ void __launch_bounds__(256) bn_apply_train_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n, int C,
                                                             const float* __restrict__ par, const float* __restrict__ saved, int relu) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % (size_t)C);
        float v = ((x[i] - saved[c]) * saved[C + c]) * par[c] + par[C + c];
        if (relu) v = fmaxf(v, 0.0f);
        y[i] = v;
    }
}

// scale = gamma / sqrt(var + eps), both steps rounded to float32 as NumPy rounds them: the hardware's v_sqrt_f32 is
// not correctly rounded, so each step is taken in double and rounded once (53 >= 2 * 24 + 2 bits: no double rounding)
__global__ void bn_scale_kernel(const float* __restrict__ par, int C, float eps, float* __restrict__ scale) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float s = (float)sqrt((double)(par[3 * C + c] + eps));
    scale[c] = (float)((double)par[c] / (double)s);
}

__global__ void __launch_bounds__(256) bn_infer_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n, int C,
                                                       const float* __restrict__ par, const float* __restrict__ scale_c, int relu) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % (size_t)C);
        const float scale = scale_c[c];
        float v = (x[i] - par[2 * C + c]) * scale + par[C + c];
        if (relu) v = fmaxf(v, 0.0f);
        y[i] = v;
    }
}

__global__ void __launch_bounds__(BN_T) bn_bwd_reduce_kernel(const float* __restrict__ x, const float* __restrict__ ymask,
                                                            const float* __restrict__ dy, size_t npx, int C,
                                                            const float* __restrict__ saved, double* __restrict__ s_dy,
                                                            double* __restrict__ s_dyx) {
    const Lane l = bn_lane(C);
    double v[2] = {0.0, 0.0};
    if (l.c >= 0) {
        const float mu = saved[l.c], is = saved[C + l.c];
        for (size_t px = (size_t)blockIdx.x * l.ppb + l.p; px < npx; px += (size_t)gridDim.x * l.ppb) {
            const size_t i = px * C + l.c;
            float g = dy[i];
            if (ymask && !(ymask[i] > 0.0f)) g = 0.0f;
            const float xh = (x[i] - mu) * is;
            v[0] += (double)g;
            v[1] += (double)g * (double)xh;
        }
    }
    double* const out[2] = {s_dy, s_dyx};
    bn_fold<2>(v, l, C, out);
}

__global__ void bn_param_grad_kernel(const double* __restrict__ s_dy, const double* __restrict__ s_dyx, int C, float* __restrict__ dgamma,
                                     float* __restrict__ dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    dbeta[c] += (float)s_dy[c];
    dgamma[c] += (float)s_dyx[c];
}

__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ ymask,
                                                           const float* __restrict__ dy, float* __restrict__ dx, size_t n, size_t npx, int C,
                                                           const float* __restrict__ par, const float* __restrict__ saved,
                                                           const double* __restrict__ s_dy, const double* __restrict__ s_dyx) {
    const float inv_n = 1.0f / (float)npx;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % (size_t)C);
        float g = dy[i];
        if (ymask && !(ymask[i] > 0.0f)) g = 0.0f;
        const float is = saved[C + c];
        const float xh = (x[i] - saved[c]) * is;
        const float m1 = (float)s_dy[c] * inv_n, m2 = (float)s_dyx[c] * inv_n;
        dx[i] += par[c] * is * (g - m1 - xh * m2);
    }
}

__device__ __forceinline__ float bf2f(uint16_t b) { return __uint_as_float((uint32_t)b << 16); }
__device__ __forceinline__ uint16_t f2bf(float f) {   // round to nearest even (finite inputs)
    uint32_t u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// bf16 tensors hold Cs = round_up(C, 8) storage channels; 8 channels (16 B) per thread, scale/shift zero on the pad channels
__global__ void __launch_bounds__(256) bn_infer_bf16_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, size_t n8, int Cs8,
                                                            const float* __restrict__ ss, int relu) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % (size_t)Cs8) * 8;
        const uint4 v = x[i];
        const uint32_t in[4] = {v.x, v.y, v.z, v.w};
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a = bf2f((uint16_t)(in[j] & 0xFFFFu)) * ss[c + 2 * j] + ss[Cs8 * 8 + c + 2 * j];
            float b = bf2f((uint16_t)(in[j] >> 16)) * ss[c + 2 * j + 1] + ss[Cs8 * 8 + c + 2 * j + 1];
            if (relu) { a = fmaxf(a, 0.0f); b = fmaxf(b, 0.0f); }
            o[j] = (uint32_t)f2bf(a) | ((uint32_t)f2bf(b) << 16);
        }
        y[i] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

inline int ew_grid(size_t n) { return (int)std::min<size_t>((n + 255) / 256, 16384); }
inline dim3 red_grid(size_t npx, int C) {
    const int cpb = C < BN_T ? C : BN_T;
    const int ppb = BN_T / cpb;
    const size_t blocks = (npx + ppb - 1) / ppb;
    return dim3((unsigned)std::min<size_t>(std::max<size_t>(blocks / 16, 1), 2048), (unsigned)cdiv(C, BN_T));
}
// the op's work buffer: [batch mean | invstd][C] float, [inference scale][C] float, then 4*C doubles of reduction scratch
inline double* scratch_of(float* saved, int C) { return (double*)((char*)saved + (((size_t)3 * C * 4 + 7) & ~(size_t)7)); }

}  // namespace

size_t bn_saved_bytes(int C) { return (((size_t)3 * C * 4 + 7) & ~(size_t)7) + (size_t)4 * C * 8; }

int bn_infer(const float* x, float* y, size_t npx, int C, const float* par, float* saved, int relu, hipStream_t st) {
    const size_t n = npx * C;
    float* scale = saved + 2 * C;
    bn_scale_kernel<<<cdiv(C, 256), 256, 0, st>>>(par, C, PSEG_BN_EPS, scale);
    bn_infer_kernel<<<ew_grid(n), 256, 0, st>>>(x, y, n, C, par, scale, relu);
    PSEG_HIP(hipGetLastError());
    return PSEG_OK;
}

int bn_train_forward(const float* x, float* y, size_t npx, int C, float* par, float* saved, int relu, int up, hipStream_t st) {
    double* sc = scratch_of(saved, C);
    PSEG_HIP(hipMemsetAsync(sc, 0, (size_t)2 * C * 8, st));
    const dim3 g = red_grid(npx, C);
    bn_sum_kernel<<<g, BN_T, 0, st>>>(x, npx, C, sc);
    bn_sqdev_kernel<<<g, BN_T, 0, st>>>(x, npx, C, sc, sc + C);
    bn_finalize_kernel<<<cdiv(C, 256), 256, 0, st>>>(sc, sc + C, npx, (double)(npx << (2 * up)), C, PSEG_BN_EPS, PSEG_BN_MOMENTUM, par, saved);
    const size_t n = npx * C;
    bn_apply_train_kernel<<<ew_grid(n), 256, 0, st>>>(x, y, n, C, par, saved, relu);
    PSEG_HIP(hipGetLastError());
    return PSEG_OK;
}

int bn_backward(const float* x, const float* y_mask, const float* dy, float* dx_accum, size_t npx, int C, const float* par,
                float* saved, float* dgamma, float* dbeta, hipStream_t st) {
    double* sc = scratch_of(saved, C) + 2 * C;
    PSEG_HIP(hipMemsetAsync(sc, 0, (size_t)2 * C * 8, st));
    bn_bwd_reduce_kernel<<<red_grid(npx, C), BN_T, 0, st>>>(x, y_mask, dy, npx, C, saved, sc, sc + C);
    bn_param_grad_kernel<<<cdiv(C, 256), 256, 0, st>>>(sc, sc + C, C, dgamma, dbeta);
    if (dx_accum) {
        const size_t n = npx * C;
        bn_bwd_apply_kernel<<<ew_grid(n), 256, 0, st>>>(x, y_mask, dy, dx_accum, n, npx, C, par, saved, sc, sc + C);
    }
    PSEG_HIP(hipGetLastError());
    return PSEG_OK;
}

int bn_infer_bf16(const uint16_t* x, uint16_t* y, size_t npx, int Cs, const float* scale_shift, int relu, hipStream_t st) {
    if (Cs % 8) return fail(PSEG_EINVAL, "bf16 tensors hold a multiple of 8 storage channels");
    const size_t n8 = npx * (size_t)(Cs / 8);
    bn_infer_bf16_kernel<<<ew_grid(n8), 256, 0, st>>>((const uint4*)x, (uint4*)y, n8, Cs / 8, scale_shift, relu);
    PSEG_HIP(hipGetLastError());
    return PSEG_OK;
}

}  // namespace pseg
