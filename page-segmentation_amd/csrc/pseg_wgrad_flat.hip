// pseg_wgrad_flat.hip -- weight gradient of the stride-1 convolutions of fcn / fcn_skip on the matrix cores, rows flattened.
//
//   dW[(tap, ci)][co] = sum over pixels p of X[p + tap][ci] * dY'[p][co]          (dY' = dY under the layer's ReLU mask)
//
// is a GEMM with M = taps x Cin rows, N = Cout columns and K = every pixel of the map.  The kernels of pseg_train.hip give
// each tap its own 16-row tiles (Cin = 20 fills 20 of 32 rows: 37 % of the matrix work multiplies padding) and let every
// wave keep ALL tiles for a quarter of the pixels (4x the accumulators, a cross-wave reduction at the end, and each tap's
// workgroup staging the same rows again).  Here the (tap, ci) pairs of KYN kernel rows are ONE row index m = tap * XC + ci,
// cut into 16-row tiles without regard to tap borders (25 x 20 = 500 rows = 32 tiles, 2 % padding), and the waves of a
// workgroup split the TILES, not the pixels: a wave keeps MW x NT accumulator tiles, walks every pixel quad of the piece,
// and nothing is reduced across waves.
//
// Data movement.  A workgroup owns a row strip x a column group and walks it column piece by column piece (PW pixels wide),
// top to bottom.  Per output row it brings ONE new X row piece (PW + KW - 1 pixels x XC channels -- contiguous in NHWC, so
// the copy is plain 16 / 8-byte vectors with no index arithmetic) into a ring of KYN + 1 row slots in LDS, and one dY row
// piece (+ its mask) into one of two buffers; the loads are issued before the multiply phase of the previous row and land
// in registers under it, so a row costs one barrier.  Image borders, the piece's right edge and rows above / below the map
// are the buffer descriptor's range check (out-of-range dwords read as zero): no compares in the copy.
// Fragments: lane (c = lane & 15, g = lane >> 4) of tile row m = (ky, kx, ci) reads X slot[ky][pixel 4q + g + kx][ci] --
// one ds_read_b32 whose address is a per-tile register + a compile-time offset per quad (XC, Cout, KW are template
// parameters for exactly that reason: the multiply phase issues no VALU address arithmetic; VALU and MFMA share a SIMD's
// issue port).  Columns >= Cout of the last dY tile read the next pixel's values: they only reach accumulator columns that
// are never written out.
//
// Summation order (pixels in walk order per strip, strips reduced by wgrad_reduce_*): float32, deterministic, not the
// oracle's order -- the train step is held to a float tolerance against torch autograd (tests/test_train_gpu.py).
#include <algorithm>

#include "pseg_wgrad.h"

namespace pseg {

typedef __attribute__((ext_vector_type(4))) float wf_f32x4;

template <int VW>
__device__ __forceinline__ void wf_bload(float* dst, __amdgpu_buffer_rsrc_t r, int voff) {
    if constexpr (VW == 4) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[k] = __builtin_bit_cast(float, (unsigned)v[k]);
    } else if constexpr (VW == 2) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, 0, 0);
        dst[0] = __builtin_bit_cast(float, (unsigned)v[0]);
        dst[1] = __builtin_bit_cast(float, (unsigned)v[1]);
    } else {
        dst[0] = __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, 0));
    }
}
__device__ __forceinline__ int itW_x(const WgradArgs& a, int mode) { return mode == 1 ? a.Wx : a.Wy; }
template <int VW>
__device__ __forceinline__ void wf_lds_store(float* dst, const float* v) {
    if constexpr (VW == 4) *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
    else if constexpr (VW == 2) *(float2*)dst = make_float2(v[0], v[1]);
    else *dst = v[0];
}

// XC / CO / KW: the layer (source channels, output channels, kernel width).  KYN: kernel rows per workgroup (1 or KW).
// NW: waves per workgroup, MW: 16-row tiles per wave (NW * MW >= tiles of KYN * KW * XC rows).  PW: pixels per piece.
template <int XC, int CO, int KW, int KYN, int NW, int MW, int PW>
__global__ __launch_bounds__(NW * 64) void wgrad_flat_kernel(WgradArgs a) {
    constexpr int NT = (CO + 15) / 16, NTHR = NW * 64;
    constexpr int QU = MW * NT >= 24 ? 2 : 4;              // quads unrolled (>= 48 independent MFMAs in flight; a full unroll hoists every fragment load and doubles the registers)
    constexpr int MROWS = KYN * KW * XC, MTILES = (MROWS + 15) / 16;
    static_assert(MTILES <= NW * MW, "tiles do not fit the waves");
    constexpr int XPX = PW + KW - 1, XROW = XPX * XC, RS = KYN + 1, YSZ = PW * CO;
    constexpr int VX = XC % 4 == 0 ? 4 : (XC % 2 == 0 ? 2 : 1), VY = CO % 4 == 0 ? 4 : (CO % 2 == 0 ? 2 : 1);
    constexpr int NXV = XROW / VX, NYV = YSZ / VY;
    constexpr int EXV = (NXV + NTHR - 1) / NTHR, EYV = (NYV + NTHR - 1) / NTHR;
    __shared__ __attribute__((aligned(16))) float Xs[RS * XROW];
    __shared__ __attribute__((aligned(16))) float Ys[2 * YSZ + 16];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p16 = lane & 15, g = lane >> 4;
    const int ky0 = blockIdx.x * KYN;
    const int strip = blockIdx.y, cgi = strip % a.cgroups, rsi = strip / a.cgroups;
    const int r0 = rsi * a.strip_rows, nrows = max(0, min(r0 + a.strip_rows, a.Hy) - r0);
    const int cpr = (a.Wy + PW - 1) / PW, cper = (cpr + a.cgroups - 1) / a.cgroups;
    const int pc0 = cgi * cper, ncols = max(0, min(pc0 + cper, cpr) - pc0);
    const int total = nrows + KYN - 1;                       // X rows staged per column piece
    const int T = nrows > 0 ? ncols * total : 0;

    // tile rows of this lane: m = (ky, kx, ci) -> float offset inside a ring slot (+ the lane's pixel g), and ky
    int aconst[MW], akyl[MW];
#pragma unroll
    for (int i = 0; i < MW; ++i) {
        const int m = min((wave * MW + i) * 16 + p16, MROWS - 1);   // (padding rows repeat the last row: never written out)
        const int tapl = m / XC, ci = m - tapl * XC, kyl = tapl / KW, kx = tapl - kyl * KW;
        aconst[i] = (kx + g) * XC + ci;
        akyl[i] = kyl;
    }
    wf_f32x4 acc[MW][NT];
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = wf_f32x4{0.f, 0.f, 0.f, 0.f};
    float bacc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) bacc[j] = 0.0f;
    const bool want_b = a.dB != nullptr && blockIdx.x == 0 && wave == 0;

    // staging registers + the byte offsets of this thread's vectors inside a row piece (-1 = none)
    float xr[EXV][VX], yr[EYV][VY], ym[EYV][VY];
    int xvo[EXV], yvo[EYV];
#pragma unroll
    for (int u = 0; u < EXV; ++u) xvo[u] = tid + u * NTHR < NXV ? (tid + u * NTHR) * VX * 4 : -1;
#pragma unroll
    for (int u = 0; u < EYV; ++u) yvo[u] = tid + u * NTHR < NYV ? (tid + u * NTHR) * VY * 4 : -1;
    const bool has_mask = a.maskY != nullptr;

    int fcol = pc0, fs = 0;                                  // the stage the next fetch() brings
    auto fetch = [&]() {
        const int x0 = fcol * PW;
        const int sy = r0 + fs + ky0 - a.pt;
        const bool rowok = sy >= 0 && sy < a.Hx;
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(a.X + (size_t)(rowok ? sy : 0) * a.xpitch * XC), 0, rowok ? a.Wx * XC * 4 : 0, 0x00020000);
        const int xb = (x0 - a.pl) * XC * 4;                 // negative at the left border: wraps past the range check -> zeros
#pragma unroll
        for (int u = 0; u < EXV; ++u) wf_bload<VX>(xr[u], xrs, xvo[u] < 0 ? -4 : xb + xvo[u]);
        if (fs >= KYN - 1) {
            const int y = r0 + fs - (KYN - 1);
            const int yb = x0 * CO * 4;
            const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.dY + (size_t)y * a.ypitch * CO), 0, a.Wy * CO * 4, 0x00020000);
#pragma unroll
            for (int u = 0; u < EYV; ++u) wf_bload<VY>(yr[u], yrs, yvo[u] < 0 ? -4 : yb + yvo[u]);
            if (has_mask) {
                const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.maskY + (size_t)y * a.ypitch * CO), 0, a.Wy * CO * 4, 0x00020000);
#pragma unroll
                for (int u = 0; u < EYV; ++u) wf_bload<VY>(ym[u], mrs, yvo[u] < 0 ? -4 : yb + yvo[u]);
            }
        }
        if (++fs == total) { fs = 0; ++fcol; }
    };

    int wslot = 0, ybuf = 0, cs = 0;
    if (T > 0) fetch();
    for (int it = 0; it < T; ++it) {
        // ---- commit the staged row(s)
        {
            float* xd = Xs + wslot * XROW;
#pragma unroll
            for (int u = 0; u < EXV; ++u)
                if (xvo[u] >= 0) {
                    if (a.in_relu)
#pragma unroll
                        for (int k = 0; k < VX; ++k) xr[u][k] = xr[u][k] > 0.0f ? xr[u][k] : 0.0f;
                    wf_lds_store<VX>(xd + (xvo[u] >> 2), xr[u]);
                }
            if (cs >= KYN - 1) {
                float* yd = Ys + ybuf * YSZ;
#pragma unroll
                for (int u = 0; u < EYV; ++u)
                    if (yvo[u] >= 0) {
                        if (has_mask)
#pragma unroll
                            for (int k = 0; k < VY; ++k) yr[u][k] = ym[u][k] > 0.0f ? yr[u][k] : 0.0f;
                        wf_lds_store<VY>(yd + (yvo[u] >> 2), yr[u]);
                    }
            }
        }
        __syncthreads();
        if (it + 1 < T) fetch();                             // lands under the multiplies below
        if (cs >= KYN - 1) {
            int sb = wslot - (KYN - 1);
            if (sb < 0) sb += RS;
            const float* xa_p[MW];
#pragma unroll
            for (int i = 0; i < MW; ++i) {
                int s = sb + akyl[i];
                if (s >= RS) s -= RS;
                xa_p[i] = Xs + s * XROW + aconst[i];
            }
            const float* yb_p = Ys + ybuf * YSZ + g * CO + p16;
#pragma unroll QU
            for (int q = 0; q < PW / 4; ++q) {
                float xa[MW], yb[NT];
#pragma unroll
                for (int j = 0; j < NT; ++j) yb[j] = yb_p[q * 4 * CO + j * 16];
#pragma unroll
                for (int i = 0; i < MW; ++i) xa[i] = xa_p[i][q * 4 * XC];
#pragma unroll
                for (int i = 0; i < MW; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[i], yb[j], acc[i][j], 0, 0, 0);
                if (want_b)
#pragma unroll
                    for (int j = 0; j < NT; ++j) bacc[j] += yb[j];
            }
            ybuf ^= 1;
        }
        wslot = wslot + 1 == RS ? 0 : wslot + 1;
        if (++cs == total) cs = 0;
    }

    // ---- D tile: lane holds rows 4g .. 4g+3, column p16 -- straight into this strip's row of `part` (or atomics)
    const int tap0 = ky0 * KW;
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = (wave * MW + i) * 16 + 4 * g + r, co = j * 16 + p16;
                if (m < MROWS && co < CO) {
                    if (a.part) a.part[(size_t)strip * a.pstride + ((size_t)tap0 * XC + m) * CO + co] = acc[i][j][r];
                    else { const int tl = m / XC; wg_out(a, strip, tap0 + tl, m - tl * XC, co, acc[i][j][r]); }
                }
            }
    if (a.dB != nullptr && blockIdx.x == 0 && wave == 0) {     // (every wave saw every pixel: wave 0 leaves the strip's four bias rows)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float v = bacc[j];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (g == 0 && j * 16 + p16 < CO) {
                wg_out_bias(a, strip, 0, j * 16 + p16, v);
                if (a.partB)
                    for (int w = 1; w < 4; ++w) wg_out_bias(a, strip, w, j * 16 + p16, 0.0f);
            }
        }
    }
}

// ---- the same walk for the WIDE layers of unet / res_unet (k3, 32 ... 1024 channels; lib/model.py:151-203, 237-307) ----------
// A launch is cut into XC x CO channel BLOCKS on blockIdx.z (64 x 64, or 32 wide where a tensor has 32 channels): a workgroup
// owns all KW x KW taps of its block -- 9 x 64 = 576 flattened rows = 36 tiles, nine per wave, no padding -- for one strip.
// The kernels of pseg_train.hip gave every TAP of a block its own workgroup, each reading X and dY again from L2 in fragment
// layout (12 global dword loads per 16 MFMAs): 0.2-0.25 of the float32 matrix peak on unet's layers.  Here the row pieces
// are staged once per workgroup for all nine taps and the multiply phase reads LDS only (13 ds_read_b32 per 36 MFMAs).
// Differences from wgrad_flat_kernel: a pixel of X / dY is a.XC / a.Cout floats apart in memory (the block's XC / CO of them
// are copied, xc0 / yc0 in), and in LDS XC + 16 / CO + 16 apart -- a pitch of 64 or 32 floats puts the four pixels g of a
// fragment read on the same banks.  Deep layers (a 32 x 24 map at 1/16 of a 512 x 384 page) get their parallelism from the
// blocks (512 -> 1024 channels: 128 of them) and one or two strips: every strip is a full copy of the layer's gradient in
// the partial-sum buffer (19 MB).
template <int XC, int CO, int KW, int NW, int MW, int PW>
__global__ __launch_bounds__(NW * 64) void wgrad_blk_kernel(WgradArgs a) {
    constexpr int NT = CO / 16, NTHR = NW * 64, KYN = KW;
    constexpr int QU = MW * NT >= 24 ? 2 : 4;
    constexpr int MROWS = KYN * KW * XC, MTILES = (MROWS + 15) / 16;
    static_assert(MTILES <= NW * MW && XC % 16 == 0 && CO % 16 == 0, "tiles do not fit the waves");
    constexpr int XP = XC + 16, YP = CO + 16;                // LDS pixel pitches
    constexpr int XPX = PW + KW - 1, XROW = XPX * XP, RS = KYN + 1, YSZ = PW * YP;
    constexpr int NXV = XPX * XC / 4, NYV = PW * CO / 4;     // float4 vectors of a row piece
    constexpr int EXV = (NXV + NTHR - 1) / NTHR, EYV = (NYV + NTHR - 1) / NTHR;
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    float* const Xs = wsm;                                   // [RS][XROW]
    float* const Ys = wsm + RS * XROW;                       // [2][YSZ]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p16 = lane & 15, g = lane >> 4;
    const int nbo = a.Cout / CO;
    const int xc0 = ((int)blockIdx.z / nbo) * XC, yc0 = ((int)blockIdx.z % nbo) * CO;
    const int strip = blockIdx.y, cgi = strip % a.cgroups, rsi = strip / a.cgroups;
    const int r0 = rsi * a.strip_rows, nrows = max(0, min(r0 + a.strip_rows, a.Hy) - r0);
    const int cpr = (a.Wy + PW - 1) / PW, cper = (cpr + a.cgroups - 1) / a.cgroups;
    const int pc0 = cgi * cper, ncols = max(0, min(pc0 + cper, cpr) - pc0);
    const int total = nrows + KYN - 1;
    const int T = nrows > 0 ? ncols * total : 0;

    int aconst[MW], akyl[MW];
#pragma unroll
    for (int i = 0; i < MW; ++i) {
        const int m = min((wave * MW + i) * 16 + p16, MROWS - 1);
        const int tapl = m / XC, ci = m - tapl * XC, kyl = tapl / KW, kx = tapl - kyl * KW;
        aconst[i] = (kx + g) * XP + ci;
        akyl[i] = kyl;
    }
    wf_f32x4 acc[MW][NT];
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = wf_f32x4{0.f, 0.f, 0.f, 0.f};
    float bacc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) bacc[j] = 0.0f;
    const bool want_b = a.dB != nullptr && xc0 == 0 && wave == 0;

    float xr[EXV][4], yr[EYV][4], ym[EYV][4];
    int xvs[EXV], xvd[EXV], yvs[EYV], yvd[EYV];              // source byte offset inside the row piece (-1: none), LDS float offset
#pragma unroll
    for (int u = 0; u < EXV; ++u) {
        const int v = tid + u * NTHR, px = v / (XC / 4), c4 = v - px * (XC / 4);
        xvs[u] = v < NXV ? (px * a.XC + xc0 + c4 * 4) * 4 : -1;
        xvd[u] = px * XP + c4 * 4;
    }
#pragma unroll
    for (int u = 0; u < EYV; ++u) {
        const int v = tid + u * NTHR, px = v / (CO / 4), c4 = v - px * (CO / 4);
        yvs[u] = v < NYV ? (px * a.Cout + yc0 + c4 * 4) * 4 : -1;
        yvd[u] = px * YP + c4 * 4;
    }
    const bool has_mask = a.maskY != nullptr;

    int fcol = pc0, fs = 0;
    auto fetch = [&]() {
        const int x0 = fcol * PW;
        const int sy = r0 + fs - a.pt;
        const bool rowok = sy >= 0 && sy < a.Hx;
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(a.X + (size_t)(rowok ? sy : 0) * a.xpitch * a.XC), 0, rowok ? a.Wx * a.XC * 4 : 0, 0x00020000);
        const int xb = (x0 - a.pl) * a.XC * 4;               // negative at the left border: wraps past the range check -> zeros
#pragma unroll
        for (int u = 0; u < EXV; ++u) wf_bload<4>(xr[u], xrs, xvs[u] < 0 ? -4 : xb + xvs[u]);
        if (fs >= KYN - 1) {
            const int y = r0 + fs - (KYN - 1);
            const int yb = x0 * a.Cout * 4;
            const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.dY + (size_t)y * a.ypitch * a.Cout), 0, a.Wy * a.Cout * 4, 0x00020000);
#pragma unroll
            for (int u = 0; u < EYV; ++u) wf_bload<4>(yr[u], yrs, yvs[u] < 0 ? -4 : yb + yvs[u]);
            if (has_mask) {
                const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.maskY + (size_t)y * a.ypitch * a.Cout), 0, a.Wy * a.Cout * 4, 0x00020000);
#pragma unroll
                for (int u = 0; u < EYV; ++u) wf_bload<4>(ym[u], mrs, yvs[u] < 0 ? -4 : yb + yvs[u]);
            }
        }
        if (++fs == total) { fs = 0; ++fcol; }
    };

    int wslot = 0, ybuf = 0, cs = 0;
    if (T > 0) fetch();
    for (int it = 0; it < T; ++it) {
        {
            float* xd = Xs + wslot * XROW;
#pragma unroll
            for (int u = 0; u < EXV; ++u)
                if (xvs[u] >= 0) {
                    if (a.in_relu)
#pragma unroll
                        for (int k = 0; k < 4; ++k) xr[u][k] = xr[u][k] > 0.0f ? xr[u][k] : 0.0f;
                    wf_lds_store<4>(xd + xvd[u], xr[u]);
                }
            if (cs >= KYN - 1) {
                float* yd = Ys + ybuf * YSZ;
#pragma unroll
                for (int u = 0; u < EYV; ++u)
                    if (yvs[u] >= 0) {
                        if (has_mask)
#pragma unroll
                            for (int k = 0; k < 4; ++k) yr[u][k] = ym[u][k] > 0.0f ? yr[u][k] : 0.0f;
                        wf_lds_store<4>(yd + yvd[u], yr[u]);
                    }
            }
        }
        __syncthreads();
        if (it + 1 < T) fetch();                             // lands under the multiplies below
        if (cs >= KYN - 1) {
            int sb = wslot - (KYN - 1);
            if (sb < 0) sb += RS;
            const float* xa_p[MW];
#pragma unroll
            for (int i = 0; i < MW; ++i) {
                int sl = sb + akyl[i];
                if (sl >= RS) sl -= RS;
                xa_p[i] = Xs + sl * XROW + aconst[i];
            }
            const float* yb_p = Ys + ybuf * YSZ + g * YP + p16;
#pragma unroll QU
            for (int q = 0; q < PW / 4; ++q) {
                float xa[MW], yb[NT];
#pragma unroll
                for (int j = 0; j < NT; ++j) yb[j] = yb_p[q * 4 * YP + j * 16];
#pragma unroll
                for (int i = 0; i < MW; ++i) xa[i] = xa_p[i][q * 4 * XP];
#pragma unroll
                for (int i = 0; i < MW; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[i], yb[j], acc[i][j], 0, 0, 0);
                if (want_b)
#pragma unroll
                    for (int j = 0; j < NT; ++j) bacc[j] += yb[j];
            }
            ybuf ^= 1;
        }
        wslot = wslot + 1 == RS ? 0 : wslot + 1;
        if (++cs == total) cs = 0;
    }

    // ---- D tile: lane holds rows 4g .. 4g+3, column p16 -- into this strip's row of `part` (or atomics)
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = (wave * MW + i) * 16 + 4 * g + r;
                if (m < MROWS) { const int tl = m / XC; wg_out(a, strip, tl, xc0 + m - tl * XC, yc0 + j * 16 + p16, acc[i][j][r]); }
            }
    if (a.dB != nullptr && xc0 == 0 && wave == 0) {            // (every wave saw every pixel: wave 0 leaves the strip's four bias rows)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float v = bacc[j];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (g == 0) {
                wg_out_bias(a, strip, 0, yc0 + j * 16 + p16, v);
                if (a.partB)
                    for (int w = 1; w < 4; ++w) wg_out_bias(a, strip, w, yc0 + j * 16 + p16, 0.0f);
            }
        }
    }
}

struct BlkInstance {
    int XC, CO, KW, NW, PW;
    int wg_per_cu;
    size_t lds;
    void (*kernel)(WgradArgs);
};
#define PSEG_BLK(XC_, CO_, KW_, NW_, MW_, PW_, OCC_) \
    {XC_, CO_, KW_, NW_, PW_, OCC_, (size_t)((KW_ + 1) * (PW_ + KW_ - 1) * (XC_ + 16) + 2 * PW_ * (CO_ + 16)) * 4, wgrad_blk_kernel<XC_, CO_, KW_, NW_, MW_, PW_>}
static const BlkInstance g_blk[] = {
    PSEG_BLK(64, 64, 3, 4, 9, 32, 1),        // 576 rows = 36 tiles: 144 accumulator registers, one wave per SIMD
    PSEG_BLK(32, 64, 3, 4, 5, 32, 2),        // 288 rows = 18 tiles (res_unet: 32-channel sources)
    PSEG_BLK(32, 32, 3, 4, 5, 32, 4),
};
#undef PSEG_BLK

bool wgrad_blk_plan(const WgradArgs& a, int taps, WgradFlatPlan* plan) {
    if (a.mode != 0 || a.stride != 1 || a.xup || taps != a.KW * a.KW || PSEG_KNOB("PSEG_WGRAD_NO_FLAT")) return false;
    if ((size_t)a.Wx * a.XC * 4 >= (1ull << 31) || (size_t)a.Wy * a.Cout * 4 >= (1ull << 31)) return false;   // 32-bit buffer offsets per row
    for (int k = 0; k < (int)(sizeof(g_blk) / sizeof(g_blk[0])); ++k) {
        const BlkInstance& f = g_blk[k];
        if (a.XC % f.XC != 0 || a.Cout % f.CO != 0 || f.KW != a.KW) continue;
        if (f.XC == 32 && a.XC % 64 == 0) continue;            // (the widest block that divides the tensor)
        if (f.CO == 32 && a.Cout % 64 == 0) continue;
        static int ncu_dev[64];
        int dev = 0;
        (void)hipGetDevice(&dev);
        dev &= 63;
        if (ncu_dev[dev] == 0) {
            int n = 0;
            ncu_dev[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
        }
        const int nz = (a.XC / f.XC) * (a.Cout / f.CO);
        const int cpr = cdiv(a.Wy, f.PW);
        // one resident round of workgroups: blocks x strips ~ what the chip holds; a strip costs a full copy of the layer's
        // gradient in the partial-sum buffer, so layers with many blocks take few
        const int target = std::max(1, ncu_dev[dev] * f.wg_per_cu / nz);
        const int min_rows = 8;
        int cgroups = cpr, rows = 1;
        for (;;) {
            const int nrs = std::max(1, target / cgroups);
            rows = cdiv(a.Hy, nrs);
            if (rows >= min_rows || cgroups == 1) break;
            cgroups = cdiv(cgroups, 2);
        }
        cgroups = cdiv(cpr, cdiv(cpr, cgroups));
        plan->instance = 1000 + k;
        plan->strip_rows = rows;
        plan->cgroups = cgroups;
        plan->nstrips = cdiv(a.Hy, rows) * cgroups;
        return true;
    }
    return false;
}

static int wgrad_blk_launch(const WgradArgs& a_in, const WgradFlatPlan& plan, hipStream_t st) {
    const BlkInstance& f = g_blk[plan.instance - 1000];
    WgradArgs a = a_in;
    a.strip_rows = plan.strip_rows;
    a.cgroups = plan.cgroups;
    static bool attr[64][sizeof(g_blk) / sizeof(g_blk[0])];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr[dev & 63][plan.instance - 1000]) {
        PSEG_HIP(hipFuncSetAttribute((const void*)f.kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)f.lds));
        attr[dev & 63][plan.instance - 1000] = true;
    }
    const dim3 grid(1, plan.nstrips, (a.XC / f.XC) * (a.Cout / f.CO));
    hipLaunchKernelGGL(f.kernel, grid, dim3(f.NW * 64), f.lds, st, a);
    PSEG_HIP(hipGetLastError());
    return PSEG_OK;
}

// ---- instance table: one fully specialised kernel per layer shape of fcn / fcn_skip (lib/model.py:50-85)
struct FlatInstance {
    int XC, CO, KW, KYN, NW, PW;
    int wg_per_cu;                                           // workgroups of this instance a CU holds (LDS / registers): sizes the grid
    void (*kernel)(WgradArgs);
};
#define PSEG_FLAT(XC_, CO_, KW_, KYN_, NW_, MW_, PW_, OCC_) {XC_, CO_, KW_, KYN_, NW_, PW_, OCC_, wgrad_flat_kernel<XC_, CO_, KW_, KYN_, NW_, MW_, PW_>}
static const FlatInstance g_flat[] = {
    PSEG_FLAT(1, 20, 5, 5, 2, 1, 64, 8),     // conv1:  25 rows (the taps) = 2 tiles; bound by the 2 x 252 MB of dY and its mask
    PSEG_FLAT(20, 30, 5, 5, 4, 8, 64, 3),    // conv2:  500 rows = 32 tiles
    PSEG_FLAT(30, 40, 5, 5, 8, 6, 32, 2),    // conv3:  750 rows = 47 tiles
    PSEG_FLAT(40, 40, 5, 5, 8, 8, 32, 2),    // conv4: 1000 rows = 63 tiles
    PSEG_FLAT(40, 60, 5, 5, 8, 8, 32, 2),    // conv5
    PSEG_FLAT(60, 60, 5, 1, 4, 5, 64, 3),    // conv6:  300 rows = 19 tiles per kernel row
    PSEG_FLAT(60, 80, 5, 1, 4, 5, 64, 2),    // conv7
    PSEG_FLAT(80, 80, 5, 1, 4, 7, 64, 2),    // deconv1 (k5 s1): 400 rows = 25 tiles
    PSEG_FLAT(60, 40, 5, 1, 4, 5, 64, 3),    // deconv3 (k5 s1), both sources of its concat
};
#undef PSEG_FLAT

bool wgrad_flat_plan(const WgradArgs& a, int taps, WgradFlatPlan* plan) {
    if (a.mode != 0 || a.stride != 1 || a.xup || taps != a.KW * a.KW || PSEG_KNOB("PSEG_WGRAD_NO_FLAT")) return false;
    if (a.KW == 3 && !PSEG_KNOB("PSEG_WGRAD_NO_BLK") && wgrad_blk_plan(a, taps, plan)) return true;   // unet / res_unet: channel blocks
    if ((size_t)a.Wx * a.XC * 4 >= (1ull << 31) || (size_t)a.Wy * a.Cout * 4 >= (1ull << 31)) return false;   // 32-bit buffer offsets per row
    for (int k = 0; k < (int)(sizeof(g_flat) / sizeof(g_flat[0])); ++k) {
        const FlatInstance& f = g_flat[k];
        if (f.XC != a.XC || f.CO != a.Cout || f.KW != a.KW) continue;
        const int kyg = f.KW / f.KYN;
        const int cpr = cdiv(a.Wy, f.PW);
        // Every workgroup does the same work per row, so the grid is sized to ONE resident round: as many workgroups as the
        // chip holds at this instance's occupancy (asked of the runtime once), never a few more -- 528 workgroups on 512
        // slots ran three rounds for the work of two (conv4: 740 us against 500).  One column piece per workgroup where the
        // map is tall enough to keep the ring primed for many rows; shorter maps give a workgroup several pieces.
        // (per device: the strip partition -- and with it the summation order of the gradient -- follows the CURRENT device's CU count
        // and occupancy; gradients are bit-reproducible per device configuration, not across different ones)
        static int wg_per_cu[64][sizeof(g_flat) / sizeof(g_flat[0])];
        static int ncu_dev[64];
        int dev = 0;
        (void)hipGetDevice(&dev);
        dev &= 63;
        if (wg_per_cu[dev][k] == 0) {
            int n = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)f.kernel, f.NW * 64, 0) != hipSuccess || n < 1) n = f.wg_per_cu;
            wg_per_cu[dev][k] = n;
        }
        if (ncu_dev[dev] == 0) {
            int n = 0;
            ncu_dev[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
        }
        const int target = ncu_dev[dev] * wg_per_cu[dev][k];
        const int min_rows = f.KYN > 1 ? 8 : 2;               // a strip shorter than this mostly primes its ring
        int cgroups = cpr, rows = 1;
        for (;;) {
            const int nrs = std::max(1, target / (cgroups * kyg));   // row strips that fit the round
            rows = cdiv(a.Hy, nrs);
            if (rows >= min_rows || cgroups == 1) break;
            cgroups = cdiv(cgroups, 2);
        }
        cgroups = cdiv(cpr, cdiv(cpr, cgroups));              // no empty group
        plan->instance = k;
        plan->strip_rows = rows;
        plan->cgroups = cgroups;
        plan->nstrips = cdiv(a.Hy, rows) * cgroups;
        return true;
    }
    return false;
}

int wgrad_flat_launch(const WgradArgs& a_in, const WgradFlatPlan& plan, hipStream_t st) {
    if (plan.instance >= 1000) return wgrad_blk_launch(a_in, plan, st);
    const FlatInstance& f = g_flat[plan.instance];
    WgradArgs a = a_in;
    a.strip_rows = plan.strip_rows;
    a.cgroups = plan.cgroups;
    const dim3 grid(f.KW / f.KYN, plan.nstrips);
    hipLaunchKernelGGL(f.kernel, grid, dim3(f.NW * 64), 0, st, a);
    PSEG_HIP(hipGetLastError());
    return PSEG_OK;
}


// ---------------------------------------------------------------------------------------------------------------------
// Two-source weight gradient of the layers whose time is the tensors they read, not their arithmetic:
//   MODE 1  Conv2DTranspose k2 s2 (lib/model.py:71,79,83):  dW[ab][ci][co] = sum over input pixels (y, x) of
//           X[y][x][ci] * dY'[2y + a][2x + b][co]  --  fcn_skip's deconv5 reads a 252 MB dY for 8.8 GFLOP;
//   MODE 0  the 1x1 logits layer (lib/model.py:85-88):      dW[ci][co] = sum over pixels of X[p][ci] * dY[p][co].
// The kernels of pseg_train.hip ran one launch per concat source (each re-reading dY) plus, for the transposed convs, a
// third pass over dY for the bias gradient: 0.35 + 0.35 + 0.33 ms for deconv5 / deconv4 / the logits layer and their biases.
// Here one launch stages, per input row piece, BOTH sources side by side ([pixel][XC0 + XC1] in LDS) and the dY rows once:
// rows m = input channel of the concatenated input (the MFMA A operand), columns n = (a, b, co) -- the two output pixels
// of an input pixel are 2 * CO contiguous floats of a dY row, so the copy is again plain vectors -- or n = co padded to one
// 16-column tile for the logits layer.  The waves split the (M tile, N tile) grid (WM x WN waves, MW x NTW tiles each); the
// bias gradient is the column sum of the same B fragments (waves with wm == 0), left as partB[strip * 4 + ab][co].
// One barrier per step: the next piece's loads are issued before the multiplies and committed to the other LDS buffer after.
template <int MODE, int XC0, int XC1, int CO, int WM, int WN, int MW, int NTW, int PW>
__global__ __launch_bounds__(WM * WN * 64) void wgrad_pair_kernel(WgradArgs a) {
    constexpr int NW = WM * WN, NTHR = NW * 64, XCT = XC0 + XC1;
    constexpr int NCOLS = MODE == 1 ? 4 * CO : 16;              // MODE 0: CO is the column PITCH of the LDS copy (16), a.Cout the live columns
    constexpr int YPX = MODE == 1 ? 2 * PW : PW;                // dY pixels per staged row
    constexpr int YROW = YPX * CO, NYR = MODE == 1 ? 2 : 1;     // floats per staged dY row, rows per step
    static_assert(WM * MW * 16 >= XCT && WN * NTW * 16 >= NCOLS, "tiles do not fit the waves");
    static_assert(MODE == 1 || CO == 16, "logits instance: 16-column LDS pitch");
    constexpr int VX = (XC0 % 4 == 0 && XC1 % 4 == 0) ? 4 : ((XC0 % 2 == 0 && XC1 % 2 == 0) ? 2 : 1);
    constexpr int VY = MODE == 1 ? (CO % 2 == 0 ? (CO % 4 == 0 ? 4 : 2) : 1) : 1;
    constexpr int NX0V = PW * XC0 / VX, NX1V = PW * XC1 / VX, NYV = MODE == 1 ? YROW / VY : PW * 16;
    constexpr int EX0 = (NX0V + NTHR - 1) / NTHR, EX1 = (NX1V + NTHR - 1) / NTHR, EY = (NYV + NTHR - 1) / NTHR;
    __shared__ __attribute__((aligned(16))) float Xs[2][PW * XCT];
    __shared__ __attribute__((aligned(16))) float Ys[2][NYR * YROW + 16];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p16 = lane & 15, g = lane >> 4;
    const int wm = wave % WM, wn = wave / WM;
    const int strip = blockIdx.x, cgi = strip % a.cgroups, rsi = strip / a.cgroups;
    const int itH = MODE == 1 ? a.Hx : a.Hy, itW = MODE == 1 ? a.Wx : a.Wy;     // the walk is over input pixels (MODE 1) / pixels (MODE 0)
    const int r0 = rsi * a.strip_rows, nrows = max(0, min(r0 + a.strip_rows, itH) - r0);
    const int cpr = (itW + PW - 1) / PW, cper = (cpr + a.cgroups - 1) / a.cgroups;
    const int pc0 = cgi * cper, ncols = max(0, min(pc0 + cper, cpr) - pc0);
    const int T = nrows * ncols;
    const int co_live = MODE == 1 ? CO : a.Cout;

    int aoff[MW], boff[NTW];
#pragma unroll
    for (int i = 0; i < MW; ++i) aoff[i] = g * XCT + min((wm * MW + i) * 16 + p16, XCT - 1);
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const int n = min((wn * NTW + j) * 16 + p16, NCOLS - 1);
        if constexpr (MODE == 1) { const int ar = n / (2 * CO); boff[j] = ar * YROW + (n - ar * 2 * CO) + g * 2 * CO; }
        else boff[j] = g * 16 + n;
    }
    wf_f32x4 acc[MW][NTW];
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[i][j] = wf_f32x4{0.f, 0.f, 0.f, 0.f};
    float bacc[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) bacc[j] = 0.0f;
    const bool want_b = a.dB != nullptr && wm == 0;
    const bool has_mask = a.maskY != nullptr;

    // this thread's vectors of a piece: source byte offset inside the row piece and destination float offset in LDS (-1: none)
    float x0r[EX0 > 0 ? EX0 : 1][VX], x1r[EX1 > 0 ? EX1 : 1][VX], yr[NYR][EY][VY], ym[NYR][EY][VY];
    int x0s[EX0 > 0 ? EX0 : 1], x0d[EX0 > 0 ? EX0 : 1], x1s[EX1 > 0 ? EX1 : 1], x1d[EX1 > 0 ? EX1 : 1], ys[EY], yd[EY];
#pragma unroll
    for (int u = 0; u < EX0; ++u) {
        const int v = tid + u * NTHR, e = v * VX, px = e / (XC0 > 0 ? XC0 : 1), c = e - px * XC0;
        x0s[u] = v < NX0V ? e * 4 : -1;
        x0d[u] = px * XCT + c;
    }
#pragma unroll
    for (int u = 0; u < EX1; ++u) {
        const int v = tid + u * NTHR, e = v * VX, px = e / (XC1 > 0 ? XC1 : 1), c = e - px * XC1;
        x1s[u] = v < NX1V ? e * 4 : -1;
        x1d[u] = px * XCT + XC0 + c;
    }
#pragma unroll
    for (int u = 0; u < EY; ++u) {
        const int v = tid + u * NTHR;
        if constexpr (MODE == 1) { ys[u] = v < NYV ? v * VY * 4 : -1; yd[u] = v * VY; }
        else { const int px = v >> 4, c = v & 15; ys[u] = (v < NYV && c < co_live) ? (px * co_live + c) * 4 : -1; yd[u] = v; }
    }
    if constexpr (MODE == 0) {     // columns >= Cout of the padded copy stay zero
        for (int i = tid; i < 2 * (YROW + 16); i += NTHR) (&Ys[0][0])[i] = 0.0f;
    }

    int frow = 0, fcol = 0;
    auto fetch = [&]() {
        const int y = r0 + frow, x0 = (pc0 + fcol) * PW;
        if constexpr (EX0 > 0) {
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(a.X + (size_t)y * a.xpitch * XC0), 0, itW_x(a, MODE) * XC0 * 4, 0x00020000);
#pragma unroll
            for (int u = 0; u < EX0; ++u) wf_bload<VX>(x0r[u], r, x0s[u] < 0 ? -4 : x0 * XC0 * 4 + x0s[u]);
        }
        if constexpr (EX1 > 0) {
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(a.X1 + (size_t)y * a.xpitch * XC1), 0, itW_x(a, MODE) * XC1 * 4, 0x00020000);
#pragma unroll
            for (int u = 0; u < EX1; ++u) wf_bload<VX>(x1r[u], r, x1s[u] < 0 ? -4 : x0 * XC1 * 4 + x1s[u]);
        }
#pragma unroll
        for (int ar = 0; ar < NYR; ++ar) {
            const int yy = MODE == 1 ? 2 * y + ar : y;
            const int xb = (MODE == 1 ? 2 * x0 : x0) * co_live * 4;
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(a.dY + (size_t)yy * a.ypitch * co_live), 0, a.Wy * co_live * 4, 0x00020000);
#pragma unroll
            for (int u = 0; u < EY; ++u) wf_bload<VY>(yr[ar][u], r, ys[u] < 0 ? -4 : xb + ys[u]);
            if (has_mask) {
                const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc((void*)(a.maskY + (size_t)yy * a.ypitch * co_live), 0, a.Wy * co_live * 4, 0x00020000);
#pragma unroll
                for (int u = 0; u < EY; ++u) wf_bload<VY>(ym[ar][u], rm, ys[u] < 0 ? -4 : xb + ys[u]);
            }
        }
        if (++fcol == ncols) { fcol = 0; ++frow; }
    };

    int buf = 0;
    if (T > 0) fetch();
    if constexpr (MODE == 0) __syncthreads();          // the zero fill is done before the first commit
    for (int it = 0; it < T; ++it) {
        {
            float* xd = Xs[buf];
#pragma unroll
            for (int u = 0; u < EX0; ++u) if (x0s[u] >= 0) wf_lds_store<VX>(xd + x0d[u], x0r[u]);
#pragma unroll
            for (int u = 0; u < EX1; ++u) if (x1s[u] >= 0) wf_lds_store<VX>(xd + x1d[u], x1r[u]);
#pragma unroll
            for (int ar = 0; ar < NYR; ++ar) {
                float* ydst = Ys[buf] + ar * YROW;
#pragma unroll
                for (int u = 0; u < EY; ++u)
                    if (ys[u] >= 0) {
                        if (has_mask)
#pragma unroll
                            for (int k = 0; k < VY; ++k) yr[ar][u][k] = ym[ar][u][k] > 0.0f ? yr[ar][u][k] : 0.0f;
                        wf_lds_store<VY>(ydst + yd[u], yr[ar][u]);
                    }
            }
        }
        __syncthreads();
        if (it + 1 < T) fetch();
        {
            const float* xb_ = Xs[buf];
            const float* yb_ = Ys[buf];
            constexpr int YQ = MODE == 1 ? 4 * 2 * CO : 4 * 16;      // floats a pixel quad advances the dY copy by
#pragma unroll 4
            for (int q = 0; q < PW / 4; ++q) {
                float xa[MW], yb[NTW];
#pragma unroll
                for (int j = 0; j < NTW; ++j) yb[j] = yb_[boff[j] + q * YQ];
#pragma unroll
                for (int i = 0; i < MW; ++i) xa[i] = xb_[aoff[i] + q * 4 * XCT];
#pragma unroll
                for (int i = 0; i < MW; ++i)
#pragma unroll
                    for (int j = 0; j < NTW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[i], yb[j], acc[i][j], 0, 0, 0);
                if (want_b)
#pragma unroll
                    for (int j = 0; j < NTW; ++j) bacc[j] += yb[j];
            }
        }
        buf ^= 1;
    }

    // ---- D tile: lane holds rows 4g .. 4g + 3 (input channel), column p16
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = (wm * MW + i) * 16 + 4 * g + r, n = (wn * NTW + j) * 16 + p16;
                if (m < XCT && n < (MODE == 1 ? NCOLS : co_live)) {
                    const int tap = MODE == 1 ? n / CO : 0, co = MODE == 1 ? n - tap * CO : n;
                    wg_out(a, strip, tap, m, co, acc[i][j][r]);
                }
            }
    if (a.dB != nullptr && wm == 0) {
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            float v = bacc[j];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            const int n = (wn * NTW + j) * 16 + p16;
            if (g == 0 && n < (MODE == 1 ? NCOLS : co_live)) {
                if constexpr (MODE == 1) { const int tap = n / CO; wg_out_bias(a, strip, tap, n - tap * CO, v); }
                else { wg_out_bias(a, strip, 0, n, v); wg_out_bias(a, strip, 1, n, 0.0f); wg_out_bias(a, strip, 2, n, 0.0f); wg_out_bias(a, strip, 3, n, 0.0f); }
            }
        }
    }
}

struct PairInstance {
    int mode, XC0, XC1, CO, NW, PW, wg_per_cu;
    void (*kernel)(WgradArgs);
};
#define PSEG_PAIR(MODE_, XC0_, XC1_, CO_, WM_, WN_, MW_, NTW_, PW_, OCC_) \
    {MODE_, XC0_, XC1_, (MODE_ == 1 ? CO_ : 0), WM_ * WN_, PW_, OCC_, wgrad_pair_kernel<MODE_, XC0_, XC1_, CO_, WM_, WN_, MW_, NTW_, PW_>}
static const PairInstance g_pair[] = {
    PSEG_PAIR(1, 30, 40, 20, 1, 5, 5, 1, 32, 3),    // fcn_skip deconv5: [deconv4, conv3] -> 20: 70 rows = 5 tiles, 80 columns = 5 tiles
    PSEG_PAIR(1, 40, 60, 30, 1, 8, 7, 1, 32, 2),    // fcn_skip deconv4: [deconv3, conv5] -> 30: 100 rows = 7 tiles, 120 columns = 8 tiles
    PSEG_PAIR(1, 80, 0, 60, 1, 8, 5, 2, 32, 2),     // deconv2: 80 -> 60: 5 x 15 tiles
    PSEG_PAIR(1, 30, 0, 20, 1, 5, 2, 1, 32, 3),     // fcn deconv5
    PSEG_PAIR(1, 40, 0, 30, 1, 8, 3, 1, 32, 2),     // fcn deconv4
    PSEG_PAIR(0, 20, 30, 16, 4, 1, 1, 1, 64, 4),    // fcn_skip logits: [deconv5, conv2] -> classes (<= 16)
    PSEG_PAIR(0, 20, 0, 16, 2, 1, 1, 1, 64, 4),     // fcn logits
};
#undef PSEG_PAIR

bool wgrad_pair_plan(const WgradArgs& a, int taps, WgradFlatPlan* plan) {
    if (a.stride != 1 || a.xup || a.in_relu || PSEG_KNOB("PSEG_WGRAD_NO_PAIR")) return false;
    if ((size_t)a.Wx * a.XC * 4 >= (1ull << 31) || (size_t)a.Wy * a.Cout * 4 >= (1ull << 31)) return false;
    for (int k = 0; k < (int)(sizeof(g_pair) / sizeof(g_pair[0])); ++k) {
        const PairInstance& f = g_pair[k];
        if (f.mode != a.mode || f.XC0 != a.XC0 || f.XC1 != a.XC - a.XC0 || (f.XC1 > 0) != (a.X1 != nullptr)) continue;
        if (f.mode == 1 ? (f.CO != a.Cout || taps != 4 || a.KW != 2) : (a.Cout > 16 || taps != 1 || a.KW != 1 || a.maskY != nullptr)) continue;
        if (a.Cin != a.XC || a.ci0 != 0) continue;
        const int itH = a.mode == 1 ? a.Hx : a.Hy, itW = a.mode == 1 ? a.Wx : a.Wy;
        static int wg_per_cu[64][sizeof(g_pair) / sizeof(g_pair[0])];       // (per device, as in wgrad_flat_plan)
        static int ncu_dev[64];
        int dev = 0;
        (void)hipGetDevice(&dev);
        dev &= 63;
        if (wg_per_cu[dev][k] == 0) {
            int n = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)f.kernel, f.NW * 64, 0) != hipSuccess || n < 1) n = f.wg_per_cu;
            wg_per_cu[dev][k] = n;
        }
        if (ncu_dev[dev] == 0) {
            int n = 0;
            ncu_dev[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
        }
        // these layers are bound by the bytes they read: a workgroup per slot of ONE resident round, whole rows per workgroup
        const int target = ncu_dev[dev] * wg_per_cu[dev][k];
        const int cpr = cdiv(itW, f.PW);
        // (row strips x column groups) with at most `target` workgroups and the fewest steps for the busiest one
        int best_cg = 1, best_rows = itH, best_steps = 1 << 30;
        for (int cg = 1; cg <= cpr; ++cg) {
            const int cper = cdiv(cpr, cg), cge = cdiv(cpr, cper);
            const int nrs = std::max(1, std::min(itH, target / cge));
            const int rows = cdiv(itH, nrs), steps = rows * cper;
            if (steps < best_steps) { best_steps = steps; best_cg = cge; best_rows = rows; }
        }
        const int cgroups = best_cg;
        plan->instance = k;
        plan->strip_rows = best_rows;
        plan->cgroups = cgroups;
        plan->nstrips = cdiv(itH, plan->strip_rows) * cgroups;
        return true;
    }
    return false;
}

int wgrad_pair_launch(const WgradArgs& a_in, const WgradFlatPlan& plan, hipStream_t st) {
    const PairInstance& f = g_pair[plan.instance];
    WgradArgs a = a_in;
    a.strip_rows = plan.strip_rows;
    a.cgroups = plan.cgroups;
    hipLaunchKernelGGL(f.kernel, dim3(plan.nstrips), dim3(f.NW * 64), 0, st, a);
    PSEG_HIP(hipGetLastError());
    return PSEG_OK;
}

}  // namespace pseg
