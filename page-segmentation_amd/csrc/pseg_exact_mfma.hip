// pseg_exact_mfma.hip -- float32-exact convolution on the matrix cores.
//
// v_mfma_f32_16x16x4_f32 accumulates its four k-values as the sequential chain
// acc = fmaf(a[k], b[k], acc), k = 0..3, and continues the chain across instructions (checked on
// MI355X by tools/microtests/mfma_f32_chain.hip: 256/256 outputs bitwise equal at K = 100).  With
// k ordered (ky, kx, ci) -- the oracle's loop order -- this kernel therefore produces the SAME
// bits as conv_exact_kernel / oracle/pseg_oracle.c, at MFMA rate (157 TFLOP/s f32 peak) instead
// of one scalar FMA chain per thread:
//   * out-of-image taps and the channel padding to a multiple of four feed x = 0 (and w = 0):
//     fmaf(0, w, acc) == acc exactly (acc is never -0: it starts at +0);
//   * then acc + bias (+ residual), ReLU -- the same operation sequence as the scalar kernel.
// Because the chain runs over ALL input channels inside each tap, the halo tile is staged in LDS
// with every input channel (float32); layers whose tile does not fit 150 KB even at two output
// rows per workgroup (unet's 1536-channel concats) stay on the scalar kernel.
//
// Layout: D[cout][pixel] per 16x16 tile; lane l = (p16 = l & 15, g = l >> 4).
//   A (weights):  lane holds w[k = 4s + g][cout = 16t + p16]   (global load, L1/L2 resident)
//   B (pixels):   lane holds x[pixel p16][channel 4s + g]        (ds_read_b32 from the LDS tile)
//   D:            lane holds couts 16t + 4g .. +3 of pixel p16  (16-byte store)
// LDS pixel stride Cp = Cin rounded up to 2 (mod 4) floats: 16 pixels x 2 lane groups hit 32
// distinct banks (conflict-free ds_read_b32 while the four lane groups stay inside one tap).
// out_sy/out_sx/out_oy/out_ox scatter the output pixel grid (Conv2DTranspose k2 s2 = four 1x1
// convolutions, one per sub-pixel, written to (2y + a, 2x + b)).
#include <algorithm>

#include "pseg_common.h"

namespace pseg {

typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int XTW = 32;   // output tile width (two 16-pixel MFMA column tiles)

// MT = pixel tiles per wave (4: two rows x two column tiles, 2: one row), NT = cout tiles per workgroup.
// FLAT = false: k-steps are (tap, four channels), channels zero-padded to a multiple of four per tap
//               (all lanes of a k-step share the tap: scalar tap loop, no per-lane bookkeeping);
// FLAT = true : K flattened across taps, k = (ky*KW + kx)*Cin + ci (first layers: Cin = 1 packs four
//               taps into one MFMA instead of wasting three quarters of it).
// Both orders are the oracle's chain order; the padded / trailing k feed w = 0 against a finite x.
template <int MT, int NT, bool FLAT>
__global__ __launch_bounds__(256) void conv_exact_mfma_kernel(ConvArgs a, int Cp, int THH, int TWH, int CB) {
    extern __shared__ __attribute__((aligned(16))) float xt[];   // [THH][TWH][Cp]
    constexpr int RW = MT / 2;            // output rows per wave
    constexpr int TH = 4 * RW;            // output rows per workgroup
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p16 = lane & 15, g = lane >> 4;
    const int tiles_x = (a.Wout + XTW - 1) / XTW;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int oy0 = ty * TH, ox0 = tx * XTW;
    const int iy0 = oy0 * a.stride - a.pt, ix0 = ox0 * a.stride - a.pl;
    const int Cin = a.C0 + a.C1;
    const int co_base = blockIdx.y * (NT * 16);
    const int Ntot = a.deconv4 ? 4 * a.Cout : a.Cout;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Channel blocks: CB == Cin (one block, the oracle's chain order) unless the caller allowed a relaxed order
    // (train step): then the chain runs (block, ky, kx, ci) over tiles of CB channels that fit LDS.
    for (int cb = 0; cb < Cin; cb += CB) {
    const int cn = min(CB, Cin - cb);
    if (cb) __syncthreads();                         // every wave is done reading the previous block's tile
    // ---- stage the halo tile: the block's input channels, zeros outside the image and in the channel pad
    const int npx = THH * TWH;
    // One item = (pixel, 64-channel slice): lanes over channels.  Eight items are fetched before the first one is used: a
    // value looked at right behind its load (ReLU, mask, the store itself) costs a full memory latency per item, and a
    // wave has ~110 of them per tile -- that chain, not the MFMAs, was most of this kernel's time on the thin layers.
    constexpr int FJ = 16;                               // 64-float pieces of a tile row in flight per wave
    if (a.dbg & 1) {
        for (int i = tid; i < npx * Cp; i += 256) xt[i] = 0.0f;
    } else if (!a.up0 && !a.up1 && cn == Cin && a.C0 < 64 && a.C1 < 64) {
        // Thin layers (every source under 64 channels: all of fcn / fcn_skip and their data gradients): a tile row is ONE
        // contiguous run of TWH * C floats per source tensor -- lanes over that run (all 64 busy instead of C of them),
        // sixteen pieces requested before the first is used; a Concatenate is two such runs side by side in the LDS pixel.
        // (The lanes-over-channels path below spent 0.85 of deconv5's 1.19 ms on its two-source tile: 0.26 TB/s.)
        for (int srcsel = 0; srcsel < (a.src1 ? 2 : 1); ++srcsel) {
        const int C = srcsel ? a.C1 : a.C0, cbase = srcsel ? a.C0 : 0;
        const float* const sbase = srcsel ? a.src1 : a.src0;
        const float* const mbase = srcsel ? nullptr : a.mask;
        const int rowf = TWH * C, nj = (rowf + 63) >> 6;
        const unsigned invc = (1u << 20) / (unsigned)C + 1u;     // e < 67 * 64: e * invc < 2^32
        int dj0[FJ];                                     // the first sixteen pieces' (pixel, LDS offset), the same for every row
#pragma unroll
        for (int j = 0; j < FJ; ++j) {
            const int e = j * 64 + lane;
            const int px = (int)(((unsigned)e * invc) >> 20), c = e - px * C;
            dj0[j] = e < rowf ? (px << 16 | (px * Cp + cbase + c)) : -1;
        }
        for (int r = wave; r < THH; r += 4) {
            const int iy = iy0 + r;
            const bool rowok = iy >= 0 && iy < a.Hin;
            const size_t ro = ((size_t)(rowok ? iy : 0) * a.Win) * C;
            const float* srow = sbase + ro + (ptrdiff_t)ix0 * C;           // (lanes left of the image are masked below)
            const float* mrow = mbase ? mbase + ro + (ptrdiff_t)ix0 * C : nullptr;
            float* drow = xt + (size_t)r * TWH * Cp;
            for (int j0 = 0; j0 < nj; j0 += FJ) {
                float v[FJ], mv[FJ];
                int dj[FJ];                              // LDS offset of the piece's float inside the tile row, or -1
#pragma unroll
                for (int j = 0; j < FJ; ++j) {
                    const int e = (j0 + j) * 64 + lane;
                    int pk = dj0[j];
                    if (j0) {
                        const int px_ = (int)(((unsigned)e * invc) >> 20), c = e - px_ * C;
                        pk = e < rowf ? (px_ << 16 | (px_ * Cp + cbase + c)) : -1;
                    }
                    const int px = pk >> 16;
                    dj[j] = pk < 0 ? -1 : (pk & 0xFFFF);
                    v[j] = 0.0f;
                    mv[j] = 1.0f;
                    if (pk >= 0 && rowok && (unsigned)(ix0 + px) < (unsigned)a.Win) {
                        v[j] = srow[e];
                        if (mrow) mv[j] = mrow[e];
                    }
                }
#pragma unroll
                for (int j = 0; j < FJ; ++j) {
                    float x = v[j];
                    if (a.in_relu) x = x > 0.0f ? x : 0.0f;
                    x = mv[j] > 0.0f ? x : 0.0f;
                    if (dj[j] >= 0) drow[dj[j]] = x;
                }
            }
        }
        }
        // the channel pad of every pixel (k-steps run over 4 * ceil(Cin / 4) channels)
        for (int i = tid; i < npx * (Cp - Cin); i += 256) {
            const int p = i / (Cp - Cin), c = Cin + i - p * (Cp - Cin);
            xt[p * Cp + c] = 0.0f;
        }
    } else {
        constexpr int SU = 8;
        const int nch = (Cp + 63) >> 6;                  // slices per pixel
        const int nit = ((npx - wave + 3) >> 2) * nch;   // items of this wave
        for (int it0 = 0; it0 < nit; it0 += SU) {
            float v[SU], mv[SU];
            int dsto[SU];
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                const int it = it0 + u;
                const int pq = it / nch, sl = it - pq * nch;
                const int p = wave + 4 * pq, cl = sl * 64 + lane;
                const int r = p / TWH, c = p - r * TWH;
                const int iy = iy0 + r, ix = ix0 + c;
                const bool in = it < nit && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
                const int ch = cb + cl;
                v[u] = 0.0f;
                mv[u] = 1.0f;
                dsto[u] = (it < nit && cl < Cp) ? p * Cp + cl : -1;
                if (in && cl < cn) {
                    if (ch < a.C0) {
                        const size_t o = ((size_t)(iy >> a.up0) * (a.Win >> a.up0) + (ix >> a.up0)) * a.C0 + ch;
                        v[u] = a.src0[o];
                        if (a.mask) mv[u] = a.mask[o];
                    } else {
                        v[u] = a.src1[((size_t)(iy >> a.up1) * (a.Win >> a.up1) + (ix >> a.up1)) * a.C1 + (ch - a.C0)];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                float x = v[u];
                if (a.in_relu) x = x > 0.0f ? x : 0.0f;
                x = mv[u] > 0.0f ? x : 0.0f;
                if (dsto[u] >= 0) xt[dsto[u]] = x;
            }
        }
    }
    __syncthreads();

    int pixoff[MT];   // float offset of this lane's pixel at tap (0,0), channel 0
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int row = wave * RW + (m >> 1), col = (m & 1) * 16 + p16;
        pixoff[m] = (row * a.stride * TWH + col * a.stride) * Cp;
    }
    // this lane's weight column per cout tile: w[k*Cout + wcol] (transposed conv k2 s2 as one GEMM over
    // n = ab*Cout + co: the sub-pixel's kernel starts ab*Cin*Cout further on), or -1 past the layer
    int wcol[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = co_base + t * 16 + p16;
        const int ab = a.deconv4 ? n / a.Cout : 0;
        wcol[t] = n < Ntot ? ab * Cin * a.Cout + (n - ab * a.Cout) : -1;
    }
    // Lanes without a weight to load (channel / cout padding, k past the layer) read the first float of the zero slack
    // every weight buffer carries behind its last element: the load is unconditional and its result needs no select, so
    // the loads of k-step s + 1 stay in flight under the MFMAs of k-step s (as a conditional load they compiled to
    // branches with a full memory wait right behind them).
    const int zoff = (a.deconv4 ? 4 : a.KH * a.KW) * Cin * a.Cout;
    float xa[MT], wa[NT], xb[MT], wb[NT];
#define PSEG_XMMA(XF, WF)                                                                        \
    _Pragma("unroll") for (int t = 0; t < NT; ++t)                                               \
        _Pragma("unroll") for (int m = 0; m < MT; ++m)                                           \
            acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(WF[t], XF[m], acc[m][t], 0, 0, 0);
    if (a.dbg & 4) {
    } else if constexpr (!FLAT) {
        const int nks = (cn + 3) >> 2;         // k-steps per tap
        for (int ky = 0; ky < a.KH; ++ky)
            for (int kx = 0; kx < a.KW; ++kx) {
                const int wbase = ((ky * a.KW + kx) * Cin + cb) * a.Cout;
                const float* wt = a.w + wbase;
                const int toff = (ky * TWH + kx) * Cp + g;
                auto load = [&](float* xf, float* wf, int s) {
                    const int ci = 4 * s + g;
#pragma unroll
                    for (int m = 0; m < MT; ++m) xf[m] = xt[pixoff[m] + toff + 4 * s];
                    const bool okc = ci < cn;
#pragma unroll
                    for (int t = 0; t < NT; ++t) wf[t] = wt[(okc && wcol[t] >= 0) ? ci * a.Cout + wcol[t] : zoff - wbase];
                };
                load(xa, wa, 0);
                int s = 0;
                for (; s + 2 <= nks; s += 2) {
                    load(xb, wb, s + 1);
                    PSEG_XMMA(xa, wa)
                    if (s + 2 < nks) load(xa, wa, s + 2);
                    PSEG_XMMA(xb, wb)
                }
                if (s < nks) { PSEG_XMMA(xa, wa) }
            }
    } else {
        const int Ktot = a.KH * a.KW * Cin;
        const int nks = (Ktot + 3) >> 2;
        const int wraps = Cin >= 4 ? 1 : 4;          // tap boundaries one step of four can cross
        int k_l = g, ci_l = g, kx_l = 0, toff_l = 0;  // this lane's k, channel, kernel column, tap offset
        auto wrap = [&]() {
            for (int w = 0; w < wraps; ++w) {
                const bool wr = ci_l >= Cin;
                ci_l -= wr ? Cin : 0;
                kx_l += wr ? 1 : 0;
                toff_l += wr ? Cp : 0;
                const bool rw = kx_l == a.KW;
                kx_l = rw ? 0 : kx_l;
                toff_l += rw ? (TWH - a.KW) * Cp : 0;
            }
        };
        wrap();
        auto load = [&](float* xf, float* wf) {
            const bool okk = k_l < Ktot;
            const int xo = okk ? toff_l + ci_l : 0;
#pragma unroll
            for (int m = 0; m < MT; ++m) xf[m] = xt[pixoff[m] + xo];
#pragma unroll
            for (int t = 0; t < NT; ++t) wf[t] = a.w[(okk && wcol[t] >= 0) ? k_l * a.Cout + wcol[t] : zoff];
            k_l += 4;
            ci_l += 4;
            wrap();
        };
        load(xa, wa);
        int s = 0;
        for (; s + 2 <= nks; s += 2) {
            load(xb, wb);
            PSEG_XMMA(xa, wa)
            load(xa, wa);
            PSEG_XMMA(xb, wb)
        }
        if (s < nks) { PSEG_XMMA(xa, wa) }
    }
    }   // channel blocks
#undef PSEG_XMMA

    // ---- epilogue: acc + bias (+ add), ReLU; lane owns n = 4g..4g+3 of pixel p16 in every tile.  The lane's four values are
    // consecutive output channels of one pixel (of one sub-pixel, for the transposed GEMM: Cout % 4 == 0 keeps a quad inside
    // its sub-pixel): one 16-byte (or two 8-byte, Cout even) store instead of four 4-byte ones -- Conv2DTranspose k2 s2 at full
    // resolution (deconv5: 252 MB of output) spent 1.28 ms in scalar scatter stores.
    const int osy = a.out_sy ? a.out_sy : 1, osx = a.out_sx ? a.out_sx : 1;
    const int pitch = a.dst_pitch ? a.dst_pitch : a.Wout;
    const int vecw = (a.Cout & 3) == 0 ? 4 : ((a.Cout & 1) == 0 ? 2 : 1);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int y = oy0 + wave * RW + (m >> 1), x = ox0 + (m & 1) * 16 + p16;
        if (y >= a.Hout || x >= a.Wout || ((a.dbg & 2) && acc[m][0][0] != 123.456f)) continue;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int n0 = co_base + t * 16 + 4 * g;
            if (n0 >= Ntot) continue;
            float v[4];
            size_t off[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + r;
                const int nn = n < Ntot ? n : n0;                 // (values past the layer are never stored)
                const int ab = a.deconv4 ? nn / a.Cout : 0;
                const int co = nn - ab * a.Cout;
                const size_t opix = a.deconv4 ? (size_t)(2 * y + (ab >> 1)) * pitch + (size_t)(2 * x + (ab & 1))
                                              : (size_t)(y * osy + a.out_oy) * pitch + (size_t)(x * osx + a.out_ox);
                off[r] = opix * a.Cout + co;
                v[r] = a.bias ? acc[m][t][r] + a.bias[co] : acc[m][t][r];
            }
            if (a.add) {
                if (vecw == 4 && n0 + 3 < Ntot) {
                    const float4 ad = *(const float4*)(a.add + off[0]);
                    v[0] = v[0] + ad.x; v[1] = v[1] + ad.y; v[2] = v[2] + ad.z; v[3] = v[3] + ad.w;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (n0 + r < Ntot) v[r] = v[r] + a.add[off[r]];
                }
            }
            if (a.relu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.0f ? v[r] : 0.0f;
            }
            if (vecw == 4 && n0 + 3 < Ntot) {
                *(float4*)(a.dst + off[0]) = make_float4(v[0], v[1], v[2], v[3]);
            } else if (vecw >= 2) {
                if (n0 + 1 < Ntot) *(float2*)(a.dst + off[0]) = make_float2(v[0], v[1]);
                else a.dst[off[0]] = v[0];
                if (n0 + 3 < Ntot) *(float2*)(a.dst + off[2]) = make_float2(v[2], v[3]);
                else if (n0 + 2 < Ntot) a.dst[off[2]] = v[2];
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n0 + r < Ntot) a.dst[off[r]] = v[r];
            }
        }
    }
}

// Layers whose all-channel halo tile does not fit LDS (unet's 1024 / 1536-channel concats, res_unet's 768 ...): the same
// MFMA formulation with BOTH operands fetched straight from global memory in fragment layout -- no tile, so no limit on the
// channel count, and the chain still runs (ky, kx, ci ascending) in one accumulator per output: the same bits as the scalar
// kernel these layers used to fall back to (2.3 TFLOP/s: unet's float32 predict spent 0.9 s per 2048x1536 page there).
//   B (pixels): lane (pixel p16, g) holds x[pixel + tap][channel 4s + g]  -- a buffer load, out-of-image taps and channels past
//               the layer read zeros through an out-of-range offset (fmaf(0, w, acc) == acc exactly);
//   A (weights): as conv_exact_mfma_kernel (zero slack for the lanes without a weight).
// A pixel's value is fetched once per tap and cout block; the re-reads hit L1 / L2 (the halo of a 4-row x 32-pixel tile).
// The fragments of k-step s + 1 are requested before the MFMAs of step s; the pre-activation ReLU is applied at the rotation.
// Sources must hold a multiple of four channels (a k-step never straddles the Concatenate).
template <int MT, int NT>
__global__ __launch_bounds__(256) void conv_exact_direct_kernel(ConvArgs a) {
    constexpr int RW = MT / 2, TH = 4 * RW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p16 = lane & 15, g = lane >> 4;
    const int tiles_x = (a.Wout + XTW - 1) / XTW;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int oy0 = ty * TH, ox0 = tx * XTW;
    const int Cin = a.C0 + a.C1;
    const int co_base = blockIdx.y * (NT * 16);
    constexpr unsigned OOB = 0xfffffff0u;
    const int H0 = a.Hin >> a.up0, W0 = a.Win >> a.up0, H1 = a.Hin >> a.up1, W1 = a.Win >> a.up1;
    const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.src0, 0, (unsigned)((size_t)H0 * W0 * a.C0 * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.src1 ? a.src1 : a.src0), 0,
                                                                        a.src1 ? (unsigned)((size_t)H1 * W1 * a.C1 * 4) : 0u, 0x00020000);
    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    int py[MT], px[MT];                    // this lane's input pixel at tap (0, 0)
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        py[m] = (oy0 + wave * RW + (m >> 1)) * a.stride - a.pt;
        px[m] = (ox0 + (m & 1) * 16 + p16) * a.stride - a.pl;
    }
    int wcol[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = co_base + t * 16 + p16;
        wcol[t] = n < a.Cout ? n : -1;
    }
    const int zoff = a.KH * a.KW * Cin * a.Cout;
    const int nks0 = a.C0 >> 2, nks = Cin >> 2;       // k-steps per tap (both sources hold multiples of four channels)
    const int total = a.KH * a.KW * nks;
    const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc((void*)(a.mask ? a.mask : a.src0), 0, a.mask ? (unsigned)((size_t)H0 * W0 * a.C0 * 4) : 0u, 0x00020000);
    float xa[MT], wa[NT], xn[MT], wn[NT], mn[MT];
    // fragments of flat step index q = tap * nks + s
    auto load = [&](int q, float* xf, float* wf, float* mf) {
        const int tap = q / nks, sidx = q - tap * nks;
        const int ky = tap / a.KW, kx = tap - ky * a.KW;
        const bool second = sidx >= nks0;                                   // wave-uniform: the step lies in src1
        const int up = second ? a.up1 : a.up0, Ws = second ? W1 : W0, C = second ? a.C1 : a.C0;
        const int c = (second ? sidx - nks0 : sidx) * 4 + g;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int iy = py[m] + ky, ix = px[m] + kx;
            const bool in = iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
            const unsigned o = in ? (unsigned)(((iy >> up) * Ws + (ix >> up)) * C + c) * 4u : OOB;
            xf[m] = __builtin_bit_cast(float, second ? __builtin_amdgcn_raw_buffer_load_b32(r1, o, 0, 0)
                                                     : __builtin_amdgcn_raw_buffer_load_b32(r0, o, 0, 0));
            // training data gradients: the value counts only where the ReLU mask (laid out as src0) is positive
            mf[m] = (a.mask && !second) ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rm, o, 0, 0)) : 1.0f;
        }
        const int ci = sidx * 4 + g;
        const int wbase = (tap * Cin + ci) * a.Cout;
#pragma unroll
        for (int t = 0; t < NT; ++t) wf[t] = a.w[wcol[t] >= 0 ? wbase + wcol[t] : zoff];
    };
    if (total > 0) load(0, xn, wn, mn);
    for (int q = 0; q < total; ++q) {
#pragma unroll
        for (int m = 0; m < MT; ++m) xa[m] = ((a.in_relu && !(xn[m] > 0.0f)) || !(mn[m] > 0.0f)) ? 0.0f : xn[m];
#pragma unroll
        for (int t = 0; t < NT; ++t) wa[t] = wn[t];
        if (q + 1 < total) load(q + 1, xn, wn, mn);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[t], xa[m], acc[m][t], 0, 0, 0);
    }
    // ---- epilogue (as conv_exact_mfma_kernel) ----
    const int osy = a.out_sy ? a.out_sy : 1, osx = a.out_sx ? a.out_sx : 1;
    const int pitch = a.dst_pitch ? a.dst_pitch : a.Wout;
    const bool vec4 = (a.Cout & 3) == 0;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int y = oy0 + wave * RW + (m >> 1), x = ox0 + (m & 1) * 16 + p16;
        if (y >= a.Hout || x >= a.Wout) continue;
        const size_t opix = (size_t)(y * osy + a.out_oy) * pitch + (size_t)(x * osx + a.out_ox);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int n0 = co_base + t * 16 + 4 * g;
            if (n0 >= a.Cout) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = n0 + r < a.Cout ? n0 + r : n0;
                v[r] = a.bias ? acc[m][t][r] + a.bias[co] : acc[m][t][r];
                if (a.add && n0 + r < a.Cout) v[r] = v[r] + a.add[opix * a.Cout + co];
                if (a.relu) v[r] = v[r] > 0.0f ? v[r] : 0.0f;
            }
            if (vec4 && n0 + 3 < a.Cout) *(float4*)(a.dst + opix * a.Cout + n0) = make_float4(v[0], v[1], v[2], v[3]);
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n0 + r < a.Cout) a.dst[opix * a.Cout + n0 + r] = v[r];
            }
        }
    }
}

template <int MT, int NT, bool FLAT>
static int launch_xm(const ConvArgs& a, int Cp, int THH, int TWH, int CB, dim3 grid, size_t lds, hipStream_t st) {
    static bool attr_set[64] = {false};
    int dev = 0;
    PSEG_HIP(hipGetDevice(&dev));
    if (!attr_set[dev & 63]) {
        PSEG_HIP(hipFuncSetAttribute((const void*)conv_exact_mfma_kernel<MT, NT, FLAT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[dev & 63] = true;
    }
    conv_exact_mfma_kernel<MT, NT, FLAT><<<grid, 256, lds, st>>>(a, Cp, THH, TWH, CB);
    PSEG_HIP(hipGetLastError());
    return PSEG_OK;
}

// Returns 1 when the layer was launched on the MFMA kernel, 0 when it does not fit (caller falls
// back to the scalar kernel), < 0 on error.
int launch_conv_exact_mfma(const ConvArgs& a_in, hipStream_t st) {
    if (PSEG_KNOB("PSEG_EXACT_SCALAR")) return 0;
    ConvArgs a = a_in;
    if (const char* dv = PSEG_DIAG_KNOB("PSEG_XM_DBG")) a.dbg = atoi(dv);   // wrong-result timing switches: diagnostic build only
    const int Cin = a.C0 + a.C1;
    if (Cin < 1 || a.Cout < 1 || a.KH != a.KW) return 0;
    const int Ntot = a.deconv4 ? 4 * a.Cout : a.Cout;
    if (Ntot < 8) return 0;                    // a handful of couts (logits): the scalar kernel wastes less
    const bool flat = Cin < 8;
    // tap-aligned k-steps read up to 4*ceil(Cin/4) channels of a pixel: the pad must be the pixel's own zeros
    int Cp = flat ? std::max(Cin, 2) : 4 * ((Cin + 3) / 4);
    while (Cp % 4 != 2) ++Cp;                  // LDS pixel stride: 2 (mod 4) floats
    const size_t budget = 150 * 1024;
    int MT = 0, THH = 0, TWH = (XTW - 1) * a.stride + a.KW;
    // 8-row tiles unless that leaves a single workgroup per CU and 4-row tiles fit twice (the MFMA pipe idles while
    // the only resident workgroup stages its tile or stores its results)
    auto lds_of = [&](int mt) { return (size_t)((4 * (mt / 2) - 1) * a.stride + a.KH) * TWH * Cp * 4; };
    const size_t l4 = lds_of(4), l2 = lds_of(2);
    int CB = Cin;
    if (l4 <= budget && (160 * 1024 / l4 >= 2 || l2 > budget || 160 * 1024 / l2 < 2 || PSEG_KNOB("PSEG_EXACT_MT4"))) MT = 4;
    else if (l2 <= budget) MT = 2;
    if (!MT && a.relaxed && !flat) {
        // the caller tolerates another summation order (train step): blocks of channels whose 4-row tile fits twice per CU
        const size_t px2 = (size_t)((4 - 1) * a.stride + a.KH) * TWH;
        CB = (int)(72 * 1024 / (px2 * 4)) / 4 * 4 - 4;
        if (CB < 16) return 0;
        Cp = CB + 2;                               // CB is a multiple of 4: stride 2 (mod 4)
        MT = 2;
    }
    // a 4-row tile that fills the CU's LDS alone (128+ channels) loses to the LDS-free kernel below: unet 97 -> 78 ms, res_unet
    // 102 -> 86 ms per float32 page (fcn_skip's 120-channel deconv3 the other way round: 7.3 vs 8.2 ms)
    if (MT == 2 && (!a.relaxed || PSEG_KNOB("PSEG_TRAIN_DIRECT")) && Cin >= 128 && !a.deconv4 && !(a.C0 & 3) && !(a.C1 & 3) && !PSEG_KNOB("PSEG_EXACT_NO_DIRECT")) MT = 0;
    if (!MT) {
        // the all-channel tile does not fit LDS: operands straight from global memory (same chain, same bits)
        if (a.deconv4 || (a.C0 & 3) || (a.C1 & 3) || PSEG_KNOB("PSEG_EXACT_NO_DIRECT")) return 0;
        if ((size_t)a.Hin * a.Win * std::max(a.C0, a.C1) * 4 >= ((size_t)1 << 32) || (size_t)a.KH * a.KW * Cin * a.Cout >= ((size_t)1 << 31)) return 0;
        const int ntall_d = cdiv(a.Cout, 16);
        const int NTd = ntall_d >= 4 ? 4 : ntall_d;
        dim3 gd(cdiv(a.Wout, XTW) * cdiv(a.Hout, 8), cdiv(ntall_d, NTd));
        switch (NTd) {
            case 1: conv_exact_direct_kernel<4, 1><<<gd, 256, 0, st>>>(a); break;
            case 2: conv_exact_direct_kernel<4, 2><<<gd, 256, 0, st>>>(a); break;
            case 3: conv_exact_direct_kernel<4, 3><<<gd, 256, 0, st>>>(a); break;
            default: conv_exact_direct_kernel<4, 4><<<gd, 256, 0, st>>>(a); break;
        }
        PSEG_HIP(hipGetLastError());
        return 1;
    }
    THH = (4 * (MT / 2) - 1) * a.stride + a.KH;
    const size_t lds = (size_t)THH * TWH * Cp * 4;
    const int ntall = cdiv(Ntot, 16);
    const int NT = ntall <= 4 ? ntall : (ntall == 5 ? 5 : 4);
    const int TH = 4 * (MT / 2);
    dim3 grid(cdiv(a.Wout, XTW) * cdiv(a.Hout, TH), cdiv(ntall, NT));
#define PSEG_XM(MT_, NT_)                                                                        \
    if (MT == MT_ && NT == NT_) {                                                                \
        if (flat) PSEG_TRY((launch_xm<MT_, NT_, true>(a, Cp, THH, TWH, CB, grid, lds, st)));         \
        else PSEG_TRY((launch_xm<MT_, NT_, false>(a, Cp, THH, TWH, CB, grid, lds, st)));             \
        return 1;                                                                                \
    }
    PSEG_XM(4, 1) PSEG_XM(4, 2) PSEG_XM(4, 3) PSEG_XM(4, 4) PSEG_XM(4, 5)
    PSEG_XM(2, 1) PSEG_XM(2, 2) PSEG_XM(2, 3) PSEG_XM(2, 4) PSEG_XM(2, 5)
#undef PSEG_XM
    return 0;
}

}  // namespace pseg
