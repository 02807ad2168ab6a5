// pseg_resize.hip -- line-height normalisation on the GPU: the pixel work of
// lib/dataset.py:114-150 (scale_binary, scale_image, prepare_images) and lib/util.py:21-29
// (preserving_resize).  The reference delegates to scikit-image 0.17.2 (resize / rescale ->
// scipy.ndimage.gaussian_filter -> warp); the kernels below restate that arithmetic in float64
// with the same operation order (no FMA contraction: -ffp-contract=off), so results equal
// oracle/resize.py bit for bit:
//   * anti-aliasing: separable Gaussian, 'mirror' boundary, scipy's symmetric accumulation order
//     (centre tap, then tap pairs from the outermost inwards); a uint8 image stays uint8 between
//     the passes (truncating cast), as scipy keeps the input dtype;
//   * warp: input coordinate = f*o + (f/2 - 0.5), f = in/out; order 0 rounds half away from zero,
//     order 3 is a 4x4 Catmull-Rom around floor(coord) (columns first, then rows), 'reflect'
//     index mapping, result clipped to the [min, max] of the (filtered) input.
// All kernels are HBM-bound streaming / gather kernels: one thread per output pixel, consecutive
// threads on consecutive x (coalesced rows); algorithmic bytes per pixel are in DESIGN.md.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "pseg_common.h"

namespace pseg {

static int rz_set_dev(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(PSEG_EHIP, "no HIP device available: the MI355X engine needs a GPU (no CPU fallback)");
    if (device < 0 || device >= n) return fail(PSEG_EINVAL, "device %d out of range (%d visible)", device, n);
    PSEG_HIP(hipSetDevice(device));
    return PSEG_OK;
}

// periodic 'mirror' / skimage 'reflect' extension (no edge repeat): d c b | a b c d | c b a
__device__ __forceinline__ int mirror_idx(int i, int n) {
    if (n == 1) return 0;
    const int p = 2 * (n - 1);
    i %= p;
    if (i < 0) i += p;
    return i >= n ? p - i : i;
}

template <typename T>
__device__ __forceinline__ double ld(const T* p, size_t i) { return (double)p[i]; }

// one pass of scipy.ndimage.correlate1d with a symmetric kernel, mode 'mirror'
template <typename T, int AXIS>
__global__ __launch_bounds__(256) void gauss_pass_kernel(const T* src, int H, int W, const double* w, int radius, T* dst) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const int i = AXIS == 0 ? y : x, n = AXIS == 0 ? H : W;
    const size_t row = (size_t)y * W;
    double acc = ld(src, row + x) * w[radius];
    for (int j = radius; j >= 1; --j) {
        const int lo = mirror_idx(i - j, n), hi = mirror_idx(i + j, n);
        const double a = AXIS == 0 ? ld(src, (size_t)lo * W + x) : ld(src, row + lo);
        const double b = AXIS == 0 ? ld(src, (size_t)hi * W + x) : ld(src, row + hi);
        acc = acc + (a + b) * w[radius - j];
    }
    dst[row + x] = (T)acc;   // uint8: C truncation, as scipy's line buffer copy
}

// order-preserving map double -> uint64 (for atomic min / max)
__device__ __forceinline__ unsigned long long ord_enc(double v) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
static inline double ord_dec(unsigned long long e) {
    const unsigned long long u = (e >> 63) ? (e & 0x7fffffffffffffffull) : ~e;
    double d;
    memcpy(&d, &u, 8);
    return d;
}

// 16-byte vector loads for the two streaming reductions below: VEC elements per thread and trip
template <typename T> struct Vec16;
template <> struct Vec16<uint8_t> { static constexpr int N = 16; };
template <> struct Vec16<double> { static constexpr int N = 2; };

// calls f(value as double) for every element of src[0..n): 16-byte loads over the aligned body, scalar tail
template <typename T, typename F>
__device__ __forceinline__ void for_each_vec(const T* src, size_t n, F f) {
    constexpr int N = Vec16<T>::N;
    const size_t nv = n / N;
    const uint4* sv = (const uint4*)src;     // hipMalloc'ed planes: 256-byte aligned
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (size_t)gridDim.x * 256) {
        const uint4 v = sv[i];
        T e[N];
        memcpy(e, &v, 16);
#pragma unroll
        for (int k = 0; k < N; ++k) f((double)e[k]);
    }
    if (blockIdx.x == 0 && threadIdx.x < n - nv * N) f((double)src[nv * N + threadIdx.x]);
}

// stats[0] = min (encoded), stats[1] = max (encoded); one atomic pair per workgroup
template <typename T>
__global__ __launch_bounds__(256) void minmax_kernel(const T* src, size_t n, unsigned long long* stats) {
    __shared__ unsigned long long smn[4], smx[4];
    double lo = 1.0e308, hi = -1.0e308;
    for_each_vec<T>(src, n, [&](double v) { lo = v < lo ? v : lo; hi = v > hi ? v : hi; });
    unsigned long long mn = ord_enc(lo), mx = ord_enc(hi);
    for (int sh = 32; sh >= 1; sh >>= 1) {
        const unsigned long long a = __shfl_xor(mn, sh), b = __shfl_xor(mx, sh);
        mn = a < mn ? a : mn;
        mx = b > mx ? b : mx;
    }
    if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) { mn = smn[w] < mn ? smn[w] : mn; mx = smx[w] > mx ? smx[w] : mx; }
        atomicMin(&stats[0], mn);
        atomicMax(&stats[1], mx);
    }
}

// stats[2] != 0  <=>  some value differs from both min and max  <=>  len(np.unique(img)) > 2
template <typename T>
__global__ __launch_bounds__(256) void third_value_kernel(const T* src, size_t n, unsigned long long* stats) {
    const unsigned long long emn = stats[0], emx = stats[1];
    const unsigned long long umn = (emn >> 63) ? (emn & 0x7fffffffffffffffull) : ~emn;
    const unsigned long long umx = (emx >> 63) ? (emx & 0x7fffffffffffffffull) : ~emx;
    const double lo = __longlong_as_double((long long)umn), hi = __longlong_as_double((long long)umx);
    bool any = false;
    for_each_vec<T>(src, n, [&](double v) { any |= (v != lo) & (v != hi); });
    if (__any(any) && (threadIdx.x & 63) == 0) atomicOr(&stats[2], 1ull);
}

__device__ __forceinline__ double cubic(double x, double f0, double f1, double f2, double f3) {
    return f1 + 0.5 * x * (f2 - f0 + x * (2.0 * f0 - 5.0 * f1 + 4.0 * f2 - f3 + x * (3.0 * (f1 - f2) + f3 - f0)));
}

// order-3 warp, output float64 clipped to [lo, hi] (encoded in stats[0..1])
template <typename T>
__global__ __launch_bounds__(256) void bicubic_kernel(const T* src, int H, int W, double* dst, int Ho, int Wo,
                                                      double fy, double ty, double fx, double tx,
                                                      const unsigned long long* stats) {
    const int ox = blockIdx.x * 256 + threadIdx.x, oy = blockIdx.y;
    if (ox >= Wo) return;
    const double yr = fy * (double)oy + ty, xc = fx * (double)ox + tx;
    const double r0f = floor(yr), c0f = floor(xc);
    const double tr = yr - r0f, tc = xc - c0f;
    const int r0 = (int)r0f - 1, c0 = (int)c0f - 1;
    int cols[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) cols[k] = mirror_idx(c0 + k, W);
    double fr[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const size_t row = (size_t)mirror_idx(r0 + k, H) * W;
        fr[k] = cubic(tc, ld(src, row + cols[0]), ld(src, row + cols[1]), ld(src, row + cols[2]), ld(src, row + cols[3]));
    }
    double v = cubic(tr, fr[0], fr[1], fr[2], fr[3]);
    unsigned long long u0 = stats[0], u1 = stats[1];
    u0 = (u0 >> 63) ? (u0 & 0x7fffffffffffffffull) : ~u0;
    u1 = (u1 >> 63) ? (u1 & 0x7fffffffffffffffull) : ~u1;
    const double lo = __longlong_as_double((long long)u0), hi = __longlong_as_double((long long)u1);
    v = v < lo ? lo : (v > hi ? hi : v);     // np.clip
    dst[(size_t)oy * Wo + ox] = v;
}

// order-0 warp = gather of whole pixels (EB bytes each)
template <int EB>
__global__ __launch_bounds__(256) void nearest_kernel(const uint8_t* src, int H, int W, uint8_t* dst, int Ho, int Wo,
                                                      double fy, double ty, double fx, double tx) {
    const int ox = blockIdx.x * 256 + threadIdx.x, oy = blockIdx.y;
    if (ox >= Wo) return;
    const double yr = fy * (double)oy + ty, xc = fx * (double)ox + tx;
    const int r = mirror_idx((int)(yr > 0.0 ? yr + 0.5 : yr - 0.5), H);
    const int c = mirror_idx((int)(xc > 0.0 ? xc + 0.5 : xc - 0.5), W);
    const uint8_t* s = src + ((size_t)r * W + c) * EB;
    uint8_t* d = dst + ((size_t)oy * Wo + ox) * EB;
    if (EB == 1) *d = *s;
    else if (EB == 2) *(uint16_t*)d = *(const uint16_t*)s;
    else if (EB == 4) *(uint32_t*)d = *(const uint32_t*)s;
    else if (EB == 8) *(uint64_t*)d = *(const uint64_t*)s;
    else
        for (int b = 0; b < EB; ++b) d[b] = s[b];
}

// elementwise steps of prepare_images (lib/dataset.py:135-146)
//   MODE 0: uint8 binary -> ink map:  out = uint8(1.0 - (gt1 ? b / 255 : b))
//   MODE 1: float64 v    -> float64:  out = 1.0 - v / 255
//   MODE 2: float64 v    -> uint8:    out = uint8((1.0 - v / 255) * 255)
//   MODE 3: float64 v    -> uint8:    out = uint8(v * 255)
template <int MODE>
__global__ __launch_bounds__(256) void prep_map_kernel(const void* src, void* dst, size_t n, const unsigned long long* stats) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (MODE == 0) {
        unsigned long long u1 = stats[1];
        u1 = (u1 >> 63) ? (u1 & 0x7fffffffffffffffull) : ~u1;
        const bool gt1 = __longlong_as_double((long long)u1) > 1.0;
        const double b = (double)((const uint8_t*)src)[i];
        ((uint8_t*)dst)[i] = (uint8_t)(1.0 - (gt1 ? b / 255.0 : b));
    } else if (MODE == 1) {
        ((double*)dst)[i] = 1.0 - ((const double*)src)[i] / 255.0;
    } else if (MODE == 2) {
        ((uint8_t*)dst)[i] = (uint8_t)((1.0 - ((const double*)src)[i] / 255.0) * 255.0);
    } else {
        ((uint8_t*)dst)[i] = (uint8_t)(((const double*)src)[i] * 255.0);
    }
}

// ---- affine warp of the augmentation pipeline (lib/data_generator.py -> keras-preprocessing
// apply_affine_transform -> scipy.ndimage.affine_transform, mode 'nearest') -------------------------------
// order 3: scipy edge-pads the plane by 12 pixels, runs the cubic B-spline prefilter (pole sqrt(3) - 2, gain 6,
// mirror initialisation) along both axes in float64 and evaluates the four-tap B-spline at
// M (r, c) + offset with coordinates clamped to the padded plane; order 0: floor(coord + 0.5), clamped.
constexpr int WARP_PAD = 12;

__global__ __launch_bounds__(256) void warp_pad_kernel(const float* src, int H, int W, double* dst, int pad) {
    const int Wp = W + 2 * pad, Hp = H + 2 * pad;
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= Wp || y >= Hp) return;
    const int sy = min(max(y - pad, 0), H - 1), sx = min(max(x - pad, 0), W - 1);
    dst[(size_t)y * Wp + x] = (double)src[(size_t)sy * W + sx];
}

// one thread = one line (row: AXIS 1, column: AXIS 0) of the padded plane, in place
// reflect = 0: mirror boundary (whole-sample symmetric: scipy's 'mirror', and what it uses for 'constant' and 'wrap');
// reflect = 1: half-sample symmetric ('reflect': c[-1 - i] = c[i]) -- scipy >= 1.6 ni_splines.c _init_causal_reflect / _anticausal_reflect
template <int AXIS>
__global__ __launch_bounds__(64) void spline3_prefilter_kernel(double* c, int Hp, int Wp, int reflect = 0) {
    const int line = blockIdx.x * 64 + threadIdx.x;
    const int nlines = AXIS == 1 ? Hp : Wp, n = AXIS == 1 ? Wp : Hp;
    if (line >= nlines) return;
    double* p = AXIS == 1 ? c + (size_t)line * Wp : c + line;
    const size_t st = AXIS == 1 ? 1 : (size_t)Wp;
    if (n == 1) return;                             // (scipy leaves a line of one sample as it is)
    const double z = -0.2679491924311227;          // sqrt(3) - 2
    for (int i = 0; i < n; ++i) p[i * st] *= 6.0;   // gain (1 - z)(1 - 1/z)
    if (reflect) {
        // c+[0] = c[0] + z sum_{i >= 0} z^i c[i] over the half-sample-symmetric extension; exact closed form for short lines
        const int hor = min(n, 28);
        const double c0 = p[0];
        double sum;
        if (hor < n) {
            double zi = 1.0;
            sum = 0.0;
            for (int i = 0; i < hor; ++i) { sum += zi * p[i * st]; zi *= z; }
            sum *= z;
        } else {
            double zn = 1.0;
            for (int i = 0; i < n; ++i) zn *= z;              // z^n
            double zi = z;
            sum = p[0] + zn * p[(size_t)(n - 1) * st];
            for (int i = 1; i < n; ++i) { sum += zi * (p[i * st] + zn * p[(size_t)(n - 1 - i) * st]); zi *= z; }
            sum *= z / (1.0 - zn * zn);
        }
        p[0] = sum + c0;
        for (int i = 1; i < n; ++i) p[i * st] += z * p[(i - 1) * st];
        p[(size_t)(n - 1) * st] *= z / (z - 1.0);
        for (int i = n - 2; i >= 0; --i) p[i * st] = z * (p[(i + 1) * st] - p[i * st]);
        return;
    }
    // causal initialisation, mirror boundary: c+[0] = sum_k z^k c[k] (|z|^k < 1e-15 after 27 terms)
    {
        const int hor = min(n, 28);
        double zi = z, sum = p[0];
        if (hor < n) {
            for (int i = 1; i < hor; ++i) { sum += zi * p[i * st]; zi *= z; }
        } else {
            const double iz = 1.0 / z;
            double z2 = 1.0;
            for (int i = 0; i < n - 1; ++i) z2 *= z;          // z^(n-1)
            double z2n = z2;
            sum = p[0] + z2 * p[(size_t)(n - 1) * st];
            z2 = z2 * z2 * iz;
            zi = z;
            for (int i = 1; i < n - 1; ++i) { sum += (zi + z2) * p[i * st]; zi *= z; z2 *= iz; }
            sum /= (1.0 - z2n * z2n);
        }
        p[0] = sum;
    }
    for (int i = 1; i < n; ++i) p[i * st] += z * p[(i - 1) * st];
    p[(size_t)(n - 1) * st] = (z / (z * z - 1.0)) * (n > 1 ? z * p[(size_t)(n - 2) * st] + p[(size_t)(n - 1) * st] : p[0] * (1.0 + z));
    for (int i = n - 2; i >= 0; --i) p[i * st] = z * (p[(i + 1) * st] - p[i * st]);
}

// fill_mode 'constant' / 'reflect' / 'wrap' (scipy.ndimage >= 1.6 geometric transforms, ni_interpolation.c map_coordinate): the plane
// is NOT padded; the prefilter runs on it with the boundary of the mode ('reflect': half-sample symmetric; 'constant' and 'wrap':
// mirror -- scipy has no exact spline boundary for those two); the coordinate is mapped by the mode ('constant': outside [0, n - 1]
// on either axis gives `cval`; 'reflect': d c b a | a b c d | d c b a; 'wrap': period n - 1, scipy's legacy 'wrap'), may stay a
// fraction outside the plane, and the tap INDICES are mapped by the prefilter's boundary.
__device__ __forceinline__ int warp_mirror(int i, int n) {
    if (n == 1) return 0;
    const int p = 2 * (n - 1);
    i = (i < 0 ? -i : i) % p;
    return i < n ? i : p - i;
}
__device__ __forceinline__ int warp_reflect_idx(int i, int n) {       // ... -2 -> 1, -1 -> 0, n -> n - 1, n + 1 -> n - 2 ...
    if (n == 1) return 0;
    const int p = 2 * n;
    i %= p;
    if (i < 0) i += p;
    return i < n ? i : p - 1 - i;
}
enum { WARP_CONST = 1, WARP_REFLECT = 2, WARP_WRAP = 3 };
template <int MODE>
__device__ __forceinline__ double warp_map_coord(double x, int n) {   // scipy map_coordinate, same operations in the same order
    if (MODE == WARP_REFLECT) {
        if (x < 0.0) {
            if (n <= 1) return 0.0;
            const double sz2 = 2.0 * n;
            if (x < -sz2) x = sz2 * (double)(long long)(-x / sz2) + x;
            x = x < -(double)n ? x + sz2 : -x - 1.0;
        } else if (x > (double)(n - 1)) {
            if (n <= 1) return 0.0;
            const double sz2 = 2.0 * n;
            x -= sz2 * (double)(long long)(x / sz2);
            if (x >= (double)n) x = sz2 - x - 1.0;
        }
    } else if (MODE == WARP_WRAP) {
        if (x < 0.0) {
            if (n <= 1) return 0.0;
            const double sz = n - 1;
            x += sz * ((double)(long long)(-x / sz) + 1.0);
        } else if (x > (double)(n - 1)) {
            if (n <= 1) return 0.0;
            const double sz = n - 1;
            x -= sz * (double)(long long)(x / sz);
        }
    }
    return x;
}
template <int ORDER, int MODE>
__global__ __launch_bounds__(256) void affine_warp_mode_kernel(const double* coef, const float* src, int H, int W, float* dst,
                                                               double m00, double m01, double m10, double m11, double o0, double o1, float cval) {
    const int c = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
    if (c >= W) return;
    double y = m00 * (double)r + m01 * (double)c + o0, x = m10 * (double)r + m11 * (double)c + o1;
    if (MODE == WARP_CONST) {
        if (y < 0.0 || y > (double)(H - 1) || x < 0.0 || x > (double)(W - 1)) { dst[(size_t)r * W + c] = cval; return; }
    } else {
        y = warp_map_coord<MODE>(y, H);
        x = warp_map_coord<MODE>(x, W);
    }
    auto tap = [](int i, int n) { return MODE == WARP_REFLECT ? warp_reflect_idx(i, n) : warp_mirror(i, n); };
    if (ORDER == 0) {
        dst[(size_t)r * W + c] = src[(size_t)tap((int)floor(y + 0.5), H) * W + tap((int)floor(x + 0.5), W)];
        return;
    }
    const int y0 = (int)floor(y), x0 = (int)floor(x);
    const double ty = y - y0, tx = x - x0;
    auto w3 = [](double t, double* w) {
        const double u = 1.0 - t;
        w[0] = u * u * u / 6.0;
        w[1] = (4.0 - 6.0 * t * t + 3.0 * t * t * t) / 6.0;
        w[2] = (4.0 - 6.0 * u * u + 3.0 * u * u * u) / 6.0;
        w[3] = t * t * t / 6.0;
    };
    double wy[4], wx[4];
    w3(ty, wy);
    w3(tx, wx);
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int yy = tap(y0 - 1 + i, H);
        double row = 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j) row += wx[j] * coef[(size_t)yy * W + tap(x0 - 1 + j, W)];
        acc += wy[i] * row;
    }
    dst[(size_t)r * W + c] = (float)acc;
}

template <int ORDER>
__global__ __launch_bounds__(256) void affine_warp_kernel(const double* coef, const float* src, int H, int W, float* dst,
                                                          double m00, double m01, double m10, double m11, double o0, double o1) {
    const int c = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
    if (c >= W) return;
    const double y = m00 * (double)r + m01 * (double)c + o0, x = m10 * (double)r + m11 * (double)c + o1;
    if (ORDER == 0) {
        const int iy = min(max((int)floor(y + 0.5), 0), H - 1), ix = min(max((int)floor(x + 0.5), 0), W - 1);
        dst[(size_t)r * W + c] = src[(size_t)iy * W + ix];
        return;
    }
    const int Hp = H + 2 * WARP_PAD, Wp = W + 2 * WARP_PAD;
    const double yp = fmin(fmax(y + WARP_PAD, 0.0), (double)(Hp - 1)), xp = fmin(fmax(x + WARP_PAD, 0.0), (double)(Wp - 1));
    const int y0 = (int)floor(yp), x0 = (int)floor(xp);
    const double ty = yp - y0, tx = xp - x0;
    auto w3 = [](double t, double* w) {
        const double u = 1.0 - t;
        w[0] = u * u * u / 6.0;
        w[1] = (4.0 - 6.0 * t * t + 3.0 * t * t * t) / 6.0;
        w[2] = (4.0 - 6.0 * u * u + 3.0 * u * u * u) / 6.0;
        w[3] = t * t * t / 6.0;
    };
    double wy[4], wx[4];
    w3(ty, wy);
    w3(tx, wx);
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int yy = min(max(y0 - 1 + i, 0), Hp - 1);
        double row = 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int xx = min(max(x0 - 1 + j, 0), Wp - 1);
            row += wx[j] * coef[(size_t)yy * Wp + xx];
        }
        acc += wy[i] * row;
    }
    dst[(size_t)r * W + c] = (float)acc;
}

static void warp_coeffs(int n_in, int n_out, double* f, double* t) {
    *f = (double)n_in / (double)n_out;
    *t = *f * 0.5 - 0.5;
}

static std::vector<double> host_gauss(double sigma, int* radius) {
    const int r = (int)(4.0 * sigma + 0.5);
    std::vector<double> w(2 * r + 1);
    double s = 0.0;
    for (int i = -r; i <= r; ++i) s += (w[i + r] = std::exp(-0.5 / (sigma * sigma) * (double)(i * i)));
    for (auto& v : w) v /= s;
    *radius = r;
    return w;   // symmetric: the [::-1] of scipy is the identity
}

struct DevMem {
    std::vector<void*> ptrs;
    ~DevMem() { for (void* p : ptrs) (void)hipFree(p); }
    template <typename T>
    int alloc(T** p, size_t count) {
        *p = nullptr;
        PSEG_HIP(hipMalloc((void**)p, std::max<size_t>(count, 1) * sizeof(T)));
        ptrs.push_back(*p);
        return PSEG_OK;
    }
};

static int stats_reset(unsigned long long* d_stats, hipStream_t st) {
    const unsigned long long init[3] = {~0ull, 0ull, 0ull};
    PSEG_HIP(hipMemcpyAsync(d_stats, init, sizeof(init), hipMemcpyHostToDevice, st));
    return PSEG_OK;
}

template <typename T>
static int compute_stats(const T* d, size_t n, unsigned long long* d_stats, bool third, hipStream_t st) {
    PSEG_TRY(stats_reset(d_stats, st));
    const int grid = (int)std::min<size_t>((n / Vec16<T>::N + 255) / 256 + 1, 1024);
    minmax_kernel<T><<<grid, 256, 0, st>>>(d, n, d_stats);
    if (third) third_value_kernel<T><<<grid, 256, 0, st>>>(d, n, d_stats);
    PSEG_HIP(hipGetLastError());
    return PSEG_OK;
}

// scale_image on device planes.  d_src: T plane (H, W); d_out: float64 (Ho, Wo).  wy / wx: host
// kernels (NULL: built here with libm exp); scratch planes are allocated from `mem`.
template <typename T>
static int scale_image_dev(DevMem& mem, const T* d_src, int H, int W, double* d_out, int Ho, int Wo,
                           const double* wy, int ry, const double* wx, int rx, unsigned long long* d_stats,
                           hipStream_t st) {
    const size_t n = (size_t)H * W;
    PSEG_TRY(compute_stats<T>(d_src, n, d_stats, true, st));
    unsigned long long hs[3];
    PSEG_HIP(hipMemcpyAsync(hs, d_stats, sizeof(hs), hipMemcpyDeviceToHost, st));
    PSEG_HIP(hipStreamSynchronize(st));
    const bool aa = hs[2] != 0;
    const T* cur = d_src;
    if (aa) {
        const double sig[2] = {std::max(0.0, ((double)H / (double)Ho - 1.0) / 2.0),
                               std::max(0.0, ((double)W / (double)Wo - 1.0) / 2.0)};
        const double* hw[2] = {wy, wx};
        int hr[2] = {ry, rx};
        for (int axis = 0; axis < 2; ++axis) {
            if (sig[axis] <= 1e-15) continue;
            std::vector<double> own;
            if (!hw[axis]) { own = host_gauss(sig[axis], &hr[axis]); hw[axis] = own.data(); }
            if (hr[axis] != (int)(4.0 * sig[axis] + 0.5))
                return fail(PSEG_EINVAL, "anti-aliasing kernel radius %d does not match sigma %.17g", hr[axis], sig[axis]);
            double* d_w = nullptr;
            T* d_tmp = nullptr;
            PSEG_TRY(mem.alloc(&d_w, (size_t)2 * hr[axis] + 1));
            PSEG_TRY(mem.alloc(&d_tmp, n));
            PSEG_HIP(hipMemcpyAsync(d_w, hw[axis], ((size_t)2 * hr[axis] + 1) * 8, hipMemcpyHostToDevice, st));
            PSEG_HIP(hipStreamSynchronize(st));   // `own` may go out of scope
            const dim3 grid(cdiv(W, 256), H);
            if (axis == 0) gauss_pass_kernel<T, 0><<<grid, 256, 0, st>>>(cur, H, W, d_w, hr[axis], d_tmp);
            else gauss_pass_kernel<T, 1><<<grid, 256, 0, st>>>(cur, H, W, d_w, hr[axis], d_tmp);
            PSEG_HIP(hipGetLastError());
            cur = d_tmp;
        }
        if (cur != d_src) PSEG_TRY(compute_stats<T>(cur, n, d_stats, false, st));   // clip range of the filtered image
    }
    double fy, ty, fx, tx;
    warp_coeffs(H, Ho, &fy, &ty);
    warp_coeffs(W, Wo, &fx, &tx);
    bicubic_kernel<T><<<dim3(cdiv(Wo, 256), Ho), 256, 0, st>>>(cur, H, W, d_out, Ho, Wo, fy, ty, fx, tx, d_stats);
    PSEG_HIP(hipGetLastError());
    return PSEG_OK;
}

static int nearest_dev(const void* d_src, int H, int W, int eb, void* d_dst, int Ho, int Wo, hipStream_t st) {
    double fy, ty, fx, tx;
    warp_coeffs(H, Ho, &fy, &ty);
    warp_coeffs(W, Wo, &fx, &tx);
    const dim3 grid(cdiv(Wo, 256), Ho);
    const uint8_t* s = (const uint8_t*)d_src;
    uint8_t* d = (uint8_t*)d_dst;
    switch (eb) {
        case 1: nearest_kernel<1><<<grid, 256, 0, st>>>(s, H, W, d, Ho, Wo, fy, ty, fx, tx); break;
        case 2: nearest_kernel<2><<<grid, 256, 0, st>>>(s, H, W, d, Ho, Wo, fy, ty, fx, tx); break;
        case 4: nearest_kernel<4><<<grid, 256, 0, st>>>(s, H, W, d, Ho, Wo, fy, ty, fx, tx); break;
        case 8: nearest_kernel<8><<<grid, 256, 0, st>>>(s, H, W, d, Ho, Wo, fy, ty, fx, tx); break;
        case 3: nearest_kernel<3><<<grid, 256, 0, st>>>(s, H, W, d, Ho, Wo, fy, ty, fx, tx); break;
        default: return fail(PSEG_EUNSUPPORTED, "element size %d (supported: 1, 2, 3, 4, 8 bytes)", eb);
    }
    PSEG_HIP(hipGetLastError());
    return PSEG_OK;
}

static int check_shape(int H, int W, int Ho, int Wo) {
    if (H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) return fail(PSEG_EINVAL, "empty image or target shape (%d,%d)->(%d,%d)", H, W, Ho, Wo);
    if ((int64_t)H * W > 0x7fffffffLL || (int64_t)Ho * Wo > 0x7fffffffLL) return fail(PSEG_EUNSUPPORTED, "image too large");
    return PSEG_OK;
}

}  // namespace pseg

using namespace pseg;

extern "C" {

int pseg_rescale_shape(int H, int W, double scale, int* Ho, int* Wo) {
    if (!Ho || !Wo) return fail(PSEG_EINVAL, "NULL argument");
    // np.round: half to even (nearbyint in the default rounding mode)
    *Ho = (int)std::nearbyint(scale * (double)H);
    *Wo = (int)std::nearbyint(scale * (double)W);
    return PSEG_OK;
}

int pseg_gaussian_kernel(double sigma, double* w, int cap, int* radius) {
    if (!radius || !(sigma > 0.0)) return fail(PSEG_EINVAL, "bad argument");
    int r = 0;
    const std::vector<double> k = host_gauss(sigma, &r);
    *radius = r;
    if (w) {
        if (cap < 2 * r + 1) return fail(PSEG_EINVAL, "kernel needs %d entries, capacity %d", 2 * r + 1, cap);
        memcpy(w, k.data(), k.size() * 8);
    }
    return PSEG_OK;
}

int pseg_resize_nearest(int device, const void* src, int H, int W, int elem_bytes, void* dst, int Ho, int Wo) {
    if (!src || !dst) return fail(PSEG_EINVAL, "NULL argument");
    PSEG_TRY(check_shape(H, W, Ho, Wo));
    PSEG_TRY(rz_set_dev(device));
    DevMem mem;
    uint8_t *d_s = nullptr, *d_d = nullptr;
    const size_t ns = (size_t)H * W * elem_bytes, nd = (size_t)Ho * Wo * elem_bytes;
    PSEG_TRY(mem.alloc(&d_s, ns));
    PSEG_TRY(mem.alloc(&d_d, nd));
    PSEG_HIP(hipMemcpy(d_s, src, ns, hipMemcpyHostToDevice));
    PSEG_TRY(nearest_dev(d_s, H, W, elem_bytes, d_d, Ho, Wo, nullptr));
    PSEG_HIP(hipMemcpy(dst, d_d, nd, hipMemcpyDeviceToHost));
    return PSEG_OK;
}

int pseg_resize_nearest_device(int device, const void* d_src, int H, int W, int elem_bytes, void* d_dst, int Ho, int Wo, void* stream) {
    if (!d_src || !d_dst) return fail(PSEG_EINVAL, "NULL argument");
    PSEG_TRY(check_shape(H, W, Ho, Wo));
    PSEG_TRY(rz_set_dev(device));
    return nearest_dev(d_src, H, W, elem_bytes, d_dst, Ho, Wo, (hipStream_t)stream);
}

int pseg_scale_image(int device, const void* src, int src_is_f64, int H, int W, double* dst, int Ho, int Wo,
                     const double* wy, int ry, const double* wx, int rx) {
    if (!src || !dst) return fail(PSEG_EINVAL, "NULL argument");
    PSEG_TRY(check_shape(H, W, Ho, Wo));
    PSEG_TRY(rz_set_dev(device));
    DevMem mem;
    const size_t n = (size_t)H * W, no = (size_t)Ho * Wo;
    unsigned long long* d_stats = nullptr;
    double* d_out = nullptr;
    PSEG_TRY(mem.alloc(&d_stats, 4));
    PSEG_TRY(mem.alloc(&d_out, no));
    if (src_is_f64) {
        double* d_s = nullptr;
        PSEG_TRY(mem.alloc(&d_s, n));
        PSEG_HIP(hipMemcpy(d_s, src, n * 8, hipMemcpyHostToDevice));
        PSEG_TRY(scale_image_dev<double>(mem, d_s, H, W, d_out, Ho, Wo, wy, ry, wx, rx, d_stats, nullptr));
    } else {
        uint8_t* d_s = nullptr;
        PSEG_TRY(mem.alloc(&d_s, n));
        PSEG_HIP(hipMemcpy(d_s, src, n, hipMemcpyHostToDevice));
        PSEG_TRY(scale_image_dev<uint8_t>(mem, d_s, H, W, d_out, Ho, Wo, wy, ry, wx, rx, d_stats, nullptr));
    }
    PSEG_HIP(hipMemcpy(dst, d_out, no * 8, hipMemcpyDeviceToHost));
    return PSEG_OK;
}

int pseg_affine_warp(int device, const float* src, int H, int W, const double m[4], const double off[2], int order,
                     float* dst) {
    return pseg_affine_warp_fill(device, src, H, W, m, off, order, 0, 0.0f, dst);
}

int pseg_affine_warp_fill(int device, const float* src, int H, int W, const double m[4], const double off[2], int order,
                          int fill_mode, float cval, float* dst) {
    if (!src || !dst || !m || !off) return fail(PSEG_EINVAL, "NULL argument");
    if (order != 0 && order != 3) return fail(PSEG_EUNSUPPORTED, "interpolation order %d (0 and 3 are built)", order);
    if (fill_mode < 0 || fill_mode > 3) return fail(PSEG_EUNSUPPORTED, "fill mode %d (0 'nearest', 1 'constant', 2 'reflect', 3 'wrap')", fill_mode);
    PSEG_TRY(check_shape(H, W, H, W));
    PSEG_TRY(rz_set_dev(device));
    DevMem mem;
    const size_t n = (size_t)H * W;
    float *d_s = nullptr, *d_d = nullptr;
    PSEG_TRY(mem.alloc(&d_s, n));
    PSEG_TRY(mem.alloc(&d_d, n));
    PSEG_HIP(hipMemcpy(d_s, src, n * 4, hipMemcpyHostToDevice));
    const dim3 grid(cdiv(W, 256), H);
    double* d_c = nullptr;
    if (order == 3) {
        const int pad = fill_mode == 0 ? WARP_PAD : 0;       // (every mode but 'nearest': scipy filters the plane itself)
        const int Hp = H + 2 * pad, Wp = W + 2 * pad;
        PSEG_TRY(mem.alloc(&d_c, (size_t)Hp * Wp));
        warp_pad_kernel<<<dim3(cdiv(Wp, 256), Hp), 256>>>(d_s, H, W, d_c, pad);
        spline3_prefilter_kernel<0><<<cdiv(Wp, 64), 64>>>(d_c, Hp, Wp, fill_mode == 2);     // axis 0 first, as scipy's spline_filter
        spline3_prefilter_kernel<1><<<cdiv(Hp, 64), 64>>>(d_c, Hp, Wp, fill_mode == 2);
    }
#define PSEG_WARP(ORD_)                                                                                                                  \
    switch (fill_mode) {                                                                                                                 \
        case 0: affine_warp_kernel<ORD_><<<grid, 256>>>(d_c, d_s, H, W, d_d, m[0], m[1], m[2], m[3], off[0], off[1]); break;             \
        case 1: affine_warp_mode_kernel<ORD_, WARP_CONST><<<grid, 256>>>(d_c, d_s, H, W, d_d, m[0], m[1], m[2], m[3], off[0], off[1], cval); break;   \
        case 2: affine_warp_mode_kernel<ORD_, WARP_REFLECT><<<grid, 256>>>(d_c, d_s, H, W, d_d, m[0], m[1], m[2], m[3], off[0], off[1], cval); break; \
        default: affine_warp_mode_kernel<ORD_, WARP_WRAP><<<grid, 256>>>(d_c, d_s, H, W, d_d, m[0], m[1], m[2], m[3], off[0], off[1], cval); break;   \
    }
    if (order == 0) { PSEG_WARP(0) } else { PSEG_WARP(3) }
#undef PSEG_WARP
    PSEG_HIP(hipGetLastError());
    PSEG_HIP(hipMemcpy(dst, d_d, n * 4, hipMemcpyDeviceToHost));
    return PSEG_OK;
}

// keras-preprocessing 1.1.2 apply_brightness_shift(x, brightness, scale=False) on one image plane (lib/trainer.py:21,33: the
// brightness_range of AugmentationSettings reaches the IMAGE generator only):
//   lo, hi = min(x), max(x); local = lo < 0 or hi > 255
//   u = uint8(local ? (x - lo) / (hi - lo) * 255 : x)                     array_to_img (float32 arithmetic, C cast = truncation)
//   v = PIL ImageEnhance.Brightness: blend(black, u, b) = b in [0, 1] ? uint8(b * u) : clip(b * u, 0, 255) truncated   (float32)
//   y = local ? v / 255 * (hi - lo) + lo : v                               img_to_array, float32
__global__ __launch_bounds__(256) void brightness_minmax_kernel(const float* x, size_t n, unsigned* mm) {
    float lo = INFINITY, hi = -INFINITY;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const float v = x[i]; lo = fminf(lo, v); hi = fmaxf(hi, v); }
    for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o)); hi = fmaxf(hi, __shfl_xor(hi, o)); }
    if ((threadIdx.x & 63) == 0) {
        // order-preserving map float -> unsigned so that atomicMin / atomicMax work on the bits
        auto key = [](float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); };
        atomicMin(&mm[0], key(lo));
        atomicMax(&mm[1], key(hi));
    }
}
__global__ __launch_bounds__(256) void brightness_apply_kernel(const float* x, float* y, size_t n, const unsigned* mm, float b) {
    auto unkey = [](unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); };
    const float lo = unkey(mm[0]), hi = unkey(mm[1]);
    const bool local = lo < 0.0f || hi > 255.0f;
    const float span = hi - lo;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float v = x[i];
        if (local) { v = v - lo; if (span != 0.0f) v = v / span; v = v * 255.0f; }
        const float u = (float)(unsigned char)(int)v;          // astype('uint8'): truncation (values are in range here)
        const float t = b * u;
        float w;
        if (b >= 0.0f && b <= 1.0f) w = (float)(unsigned char)(int)t;
        else w = t <= 0.0f ? 0.0f : (t >= 255.0f ? 255.0f : (float)(unsigned char)(int)t);
        y[i] = local ? w / 255.0f * span + lo : w;
    }
}

int pseg_brightness_shift(int device, const float* src, int64_t n, float brightness, float* dst) {
    if (!src || !dst || n < 1) return fail(PSEG_EINVAL, "NULL argument / empty plane");
    PSEG_TRY(rz_set_dev(device));
    DevMem mem;
    float *d_s = nullptr, *d_d = nullptr;
    unsigned* d_mm = nullptr;
    PSEG_TRY(mem.alloc(&d_s, (size_t)n));
    PSEG_TRY(mem.alloc(&d_d, (size_t)n));
    PSEG_TRY(mem.alloc(&d_mm, 2));
    const unsigned init[2] = {0xffffffffu, 0u};
    PSEG_HIP(hipMemcpy(d_mm, init, 8, hipMemcpyHostToDevice));
    PSEG_HIP(hipMemcpy(d_s, src, (size_t)n * 4, hipMemcpyHostToDevice));
    const int grid = (int)std::min<size_t>(((size_t)n + 255) / 256, 2048);
    brightness_minmax_kernel<<<grid, 256>>>(d_s, (size_t)n, d_mm);
    brightness_apply_kernel<<<grid, 256>>>(d_s, d_d, (size_t)n, d_mm, brightness);
    PSEG_HIP(hipGetLastError());
    PSEG_HIP(hipMemcpy(dst, d_d, (size_t)n * 4, hipMemcpyDeviceToHost));
    return PSEG_OK;
}

int pseg_prepare_images(int device, const uint8_t* image, const uint8_t* binary, int H0, int W0, int H1, int W1,
                        const double* wy1, int ry1, const double* wx1, int rx1, int H2, int W2,
                        const double* wy2, int ry2, const double* wx2, int rx2, uint8_t* out_img,
                        uint8_t* out_bin, uint8_t* out_orig_bin, double* out_stage1) {
    if (!image || !binary || !out_img || !out_bin) return fail(PSEG_EINVAL, "NULL argument");
    PSEG_TRY(check_shape(H0, W0, H1, W1));
    const bool two = H2 > 0 && W2 > 0;
    if (two) PSEG_TRY(check_shape(H1, W1, H2, W2));
    PSEG_TRY(rz_set_dev(device));
    DevMem mem;
    hipStream_t st = nullptr;
    const size_t n0 = (size_t)H0 * W0, n1 = (size_t)H1 * W1, n2 = two ? (size_t)H2 * W2 : 0;
    uint8_t *d_img = nullptr, *d_bin = nullptr, *d_ink0 = nullptr, *d_ink1 = nullptr, *d_o8 = nullptr;
    double* d_s1 = nullptr;
    unsigned long long* d_stats = nullptr;
    PSEG_TRY(mem.alloc(&d_img, n0));
    PSEG_TRY(mem.alloc(&d_bin, n0));
    PSEG_TRY(mem.alloc(&d_ink0, n0));
    PSEG_TRY(mem.alloc(&d_ink1, n1));
    PSEG_TRY(mem.alloc(&d_s1, n1));
    PSEG_TRY(mem.alloc(&d_o8, two ? n2 : n1));
    PSEG_TRY(mem.alloc(&d_stats, 4));
    PSEG_HIP(hipMemcpyAsync(d_img, image, n0, hipMemcpyHostToDevice, st));
    PSEG_HIP(hipMemcpyAsync(d_bin, binary, n0, hipMemcpyHostToDevice, st));
    // binary: orig_bin = b/255 if max > 1 else b; ink = uint8(1 - orig_bin); bin = 1 - nearest(orig_bin)
    // (the gather commutes with the per-pixel map, so the ink map is gathered)
    PSEG_TRY(compute_stats<uint8_t>(d_bin, n0, d_stats, false, st));
    prep_map_kernel<0><<<(unsigned)((n0 + 255) / 256), 256, 0, st>>>(d_bin, d_ink0, n0, d_stats);
    PSEG_HIP(hipGetLastError());
    PSEG_TRY(nearest_dev(d_ink0, H0, W0, 1, d_ink1, H1, W1, st));
    // image: stage 1 on the uint8 scan
    PSEG_TRY(scale_image_dev<uint8_t>(mem, d_img, H0, W0, d_s1, H1, W1, wy1, ry1, wx1, rx1, d_stats, st));
    if (out_stage1) PSEG_HIP(hipMemcpyAsync(out_stage1, d_s1, n1 * 8, hipMemcpyDeviceToHost, st));
    if (!two) {
        prep_map_kernel<2><<<(unsigned)((n1 + 255) / 256), 256, 0, st>>>(d_s1, d_o8, n1, d_stats);
        PSEG_HIP(hipGetLastError());
        PSEG_HIP(hipMemcpyAsync(out_img, d_o8, n1, hipMemcpyDeviceToHost, st));
        PSEG_HIP(hipMemcpyAsync(out_bin, d_ink1, n1, hipMemcpyDeviceToHost, st));
    } else {
        double *d_f1 = nullptr, *d_s2 = nullptr;
        uint8_t* d_ink2 = nullptr;
        PSEG_TRY(mem.alloc(&d_f1, n1));
        PSEG_TRY(mem.alloc(&d_s2, n2));
        PSEG_TRY(mem.alloc(&d_ink2, n2));
        prep_map_kernel<1><<<(unsigned)((n1 + 255) / 256), 256, 0, st>>>(d_s1, d_f1, n1, d_stats);
        PSEG_HIP(hipGetLastError());
        PSEG_TRY(scale_image_dev<double>(mem, d_f1, H1, W1, d_s2, H2, W2, wy2, ry2, wx2, rx2, d_stats, st));
        prep_map_kernel<3><<<(unsigned)((n2 + 255) / 256), 256, 0, st>>>(d_s2, d_o8, n2, d_stats);
        PSEG_HIP(hipGetLastError());
        PSEG_TRY(nearest_dev(d_ink1, H1, W1, 1, d_ink2, H2, W2, st));
        PSEG_HIP(hipMemcpyAsync(out_img, d_o8, n2, hipMemcpyDeviceToHost, st));
        PSEG_HIP(hipMemcpyAsync(out_bin, d_ink2, n2, hipMemcpyDeviceToHost, st));
    }
    if (out_orig_bin) PSEG_HIP(hipMemcpyAsync(out_orig_bin, d_ink0, n0, hipMemcpyDeviceToHost, st));
    PSEG_HIP(hipStreamSynchronize(st));
    return PSEG_OK;
}

}  // extern "C"
