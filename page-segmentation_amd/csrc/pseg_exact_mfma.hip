// pseg_exact_mfma.hip -- float32-exact convolution on the matrix cores.
//
// v_mfma_f32_16x16x4_f32 accumulates its four k-values as the sequential chain
// acc = fmaf(a[k], b[k], acc), k = 0..3, and continues the chain across instructions (checked on
// MI355X by tools/microtests/mfma_f32_chain.hip: 256/256 outputs bitwise equal at K = 100).  With
// k ordered as the oracle's loop nest -- (block of 16 input channels, ky, kx, ci inside the block), oracle/pseg_oracle.c
// -- the kernels below therefore produce the SAME bits as the scalar chain, at MFMA rate (157 TFLOP/s f32 peak):
//   * out-of-image taps and the channel padding of a block to a multiple of four feed x = 0 (and w = 0):
//     fmaf(0, w, acc) == acc exactly (acc is never -0: it starts at +0);
//   * then acc + bias (+ residual), ReLU -- the same operation sequence as the scalar kernel.
// Round 3: the chain is BLOCKED over the input channels (rounds 1-2 ran it over all channels inside each tap, which
// forced all-channel LDS tiles: 140 KB for fcn_skip's 120-channel layer, no LDS form at all for unet's 1024).  A
// workgroup now stages a 16-channel slab of its halo tile (31 KB for a k5 layer: four to five workgroups per CU, so one
// workgroup's staging hides under the others' MFMAs), runs every tap over it, and moves to the next slab with the
// accumulators in registers; any channel count fits, predict and train share one order and one kernel.
//
// Layout: D[cout][pixel] per 16x16 tile; lane l = (p16 = l & 15, g = l >> 4).
//   A (weights):  lane holds w[k = 4s + g][cout = 16t + p16]   (global load, L1/L2 resident)
//   B (pixels):   lane holds x[pixel p16][channel 4s + g]        (ds_read_b32 from the LDS tile)
//   D:            lane holds couts 16t + 4g .. +3 of pixel p16  (16-byte store)
// LDS pixel stride 18 floats (2 mod 4): 16 pixels x 2 lane groups hit 32 distinct banks (conflict-free ds_read_b32).
// out_sy/out_sx/out_oy/out_ox scatter the output pixel grid (Conv2DTranspose k2 s2 = four 1x1
// convolutions, one per sub-pixel, written to (2y + a, 2x + b)).
#include <algorithm>
#include <type_traits>

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

// ---- blocked-chain kernel (every layer with Cin >= 8) -----------------------------------------------------------------
constexpr int XCB = PSEG_CHAIN_BLOCK;   // input channels per pass of the chain (oracle/pseg_oracle.c: ORC_CHAIN_BLOCK)
constexpr int XCP = XCB + 2;     // LDS pixel pitch in floats

// MT = pixel tiles per wave (4: two rows x two column tiles, 2: one row), NT = cout tiles per workgroup (<= 4: 64 accumulator
// registers, four waves per SIMD).  One (tap, four channels) k-step = MT ds_read_b32 + NT global loads for MT x NT MFMAs of
// 32 cycles each; the k-steps of a slab form ONE software-pipelined stream across taps (the fragments of step q + 1 are
// requested before the MFMAs of step q; a per-tap pipeline would expose a load latency every four k-steps).
// REM (0, 4 or 8): output channels left over behind the NT full tiles.  A 16-row tile for 4 or 8 channels multiplies
// padding (Cout = 40 on three tiles: 17 %, Cout = 20 on two: 37 %).  The left-over rows are instead filled with the same
// channels of DX = 16 / REM horizontally neighbouring pixels: row (dx, j) of the tile is channel 16 NT + j of pixel
// DX * p + dx, its columns are pixel groups p, and its kernel is the layer's kernel shifted by dx along kx (zero outside:
// KW + DX - 1 taps per kernel row; a.wrem, built by wrem_kernel).  Each output still sees its own taps in (slab, ky, kx, ci)
// order with exact-zero products in between -- the same bits -- and the tile covers DX times the pixels: 6/5 x 1/2 (REM 8)
// or 8/5 x 1/4 (REM 4) of a padded tile's MFMAs.
template <int MT, int NT, int TAILK, int REM = 0>
__global__ __launch_bounds__(256) void conv_xb_kernel(ConvArgs a, int THH, int TWH, unsigned inv_twh) {
    extern __shared__ __attribute__((aligned(16))) float xt[];   // [THH][TWH][XCP]
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
    const int npx = THH * TWH;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    int xbase[MT];   // float offset of this lane's fragment element at tap (0,0), k-step 0 of a slab: its pixel, channel g
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int row = wave * RW + (m >> 1), col = (m & 1) * 16 + p16;
        xbase[m] = (row * a.stride * TWH + col * a.stride) * XCP + g;
    }
    // this lane's weight byte offset per cout tile: w[(k = g)*Cout + column] (transposed conv k2 s2 as one GEMM over
    // n = ab*Cout + co: the sub-pixel's kernel starts ab*Cin*Cout further on); out of range past the layer (reads 0)
    const unsigned wbytes = (unsigned)((size_t)(a.deconv4 ? 4 : a.KH * a.KW) * Cin * a.Cout * 4);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, wbytes, 0x00020000);
    unsigned voff[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = co_base + t * 16 + p16;
        const int ab = a.deconv4 ? n / a.Cout : 0;
        voff[t] = n < Ntot ? (unsigned)(ab * Cin * a.Cout + (n - ab * a.Cout) + g * a.Cout) * 4u : 0x7ffffff0u;
    }
    const bool pairs = ((a.C0 | a.C1) & 1) == 0;     // even channel counts: a lane stages two channels with 8-byte accesses
    // left-over channels (REM): RT tiles per wave over its RW rows -- pixel pairs of one row each (REM 8) or pixel quads of
    // both rows (REM 4: columns 0-7 the upper row's quads, 8-15 the lower row's)
    constexpr int DX = REM ? 16 / REM : 1;
    constexpr int RT = REM == 8 ? RW : (REM == 4 ? RW / 2 : 0);
    static_assert(REM == 0 || (MT == 4 && (REM == 4 || REM == 8)), "left-over channel tiles: 8-row workgroup tiles only");
    f32x4 accr[RT ? RT : 1];
    int xrbase[RT ? RT : 1];
#pragma unroll
    for (int rt = 0; rt < (RT ? RT : 1); ++rt) {
        accr[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int row = REM == 8 ? wave * RW + rt : wave * RW + rt * 2 + (p16 >> 3);
        const int col = REM == 8 ? 2 * p16 : 4 * (p16 & 7);
        xrbase[rt] = (row * TWH + col) * XCP + g;
    }
    const int KWR = a.KW + DX - 1;
    const __amdgpu_buffer_rsrc_t wrrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(REM ? a.wrem : a.w), 0, REM ? (unsigned)((size_t)a.KH * KWR * Cin * 16 * 4) : 0u, 0x00020000);
    const unsigned voffr = (unsigned)(g * 16 + p16) * 4u;

    // One slab = stage 16 channels of the halo tile, run every tap over them.  `nks_tag` carries the k-steps per tap as a
    // compile-time constant.  The full slabs run in the loop below (four k-steps per tap); a shorter last slab (TAILK, picked
    // by the host from the channel count) follows the loop as straight-line code -- with both forms inside one loop the
    // accumulators were allocated more than twice over (80 AGPRs for 32 accumulator registers at NT = 2, 156 for 64 at
    // NT = 4: two waves per SIMD instead of three or four).
    auto process_slab = [&](const int cb, auto nks_tag) {
        const int cn = min(XCB, Cin - cb);
        if (cb) __syncthreads();                     // every wave is done reading the previous slab
        // ---- stage the slab: channels cb .. cb+15 of the halo tile (zeros outside the image and past the layer's channels).
        // An item = (pixel, channel pair): eight pixels per wave trip, their 16 channels 64 contiguous bytes each; SU trips are
        // requested before the first value is touched (a value looked at right behind its load costs a full memory latency).
        constexpr int SU = 8;
        if (PSEG_DIAG && (a.dbg & 1)) {                 // timing ablation (diagnostic build): no staging loads
            for (int i = tid; i < npx * XCP; i += 256) xt[i] = 0.0f;
        } else if (pairs && !a.mask) {
            // the common form (forward layers, data gradients without a ReLU mask): no mask values to keep in flight
            const int sp = lane >> 3, c2 = (lane & 7) * 2;
            const int ch = cb + c2;
            const bool chok = ch < Cin;
            const bool second = ch >= a.C0;
            const float* const sb = second ? a.src1 : a.src0;
            const int C = second ? a.C1 : a.C0, chl = second ? ch - a.C0 : ch, up = second ? a.up1 : a.up0;
            const int Ws = a.Win >> up;
            const int ngrp = (npx + 7) >> 3;
            for (int g0 = wave; g0 < ngrp; g0 += 4 * SU) {
                float2 v[SU];
                int dsto[SU];
#pragma unroll
                for (int u = 0; u < SU; ++u) {
                    const int grp = g0 + 4 * u, p = grp * 8 + sp;
                    const bool valid = grp < ngrp && p < npx;
                    const int r = (int)(((unsigned)p * inv_twh) >> 20), c = p - r * TWH;
                    const int iy = iy0 + r, ix = ix0 + c;
                    dsto[u] = valid ? p * XCP + c2 : -1;
                    v[u] = make_float2(0.0f, 0.0f);
                    if (valid && chok && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win)
                        v[u] = *(const float2*)(sb + ((size_t)(iy >> up) * Ws + (ix >> up)) * C + chl);
                }
#pragma unroll
                for (int u = 0; u < SU; ++u) {
                    float2 x = v[u];
                    if (a.in_relu) { x.x = x.x > 0.0f ? x.x : 0.0f; x.y = x.y > 0.0f ? x.y : 0.0f; }
                    if (dsto[u] >= 0) *(float2*)(xt + dsto[u]) = x;
                }
            }
        } else if (pairs) {
            const int sp = lane >> 3, c2 = (lane & 7) * 2;
            const int ch = cb + c2;
            const bool chok = ch < Cin;
            const bool second = ch >= a.C0;
            const float* const sb = second ? a.src1 : a.src0;
            const int C = second ? a.C1 : a.C0, chl = second ? ch - a.C0 : ch, up = second ? a.up1 : a.up0;
            const int Ws = a.Win >> up;
            const float* const mb = second ? nullptr : a.mask;
            const int ngrp = (npx + 7) >> 3;
            constexpr int SM = 4;
            for (int g0 = wave; g0 < ngrp; g0 += 4 * SM) {
                float2 v[SM], mv[SM];
                int dsto[SM];
#pragma unroll
                for (int u = 0; u < SM; ++u) {
                    const int grp = g0 + 4 * u, p = grp * 8 + sp;
                    const bool valid = grp < ngrp && p < npx;
                    const int r = (int)(((unsigned)p * inv_twh) >> 20), c = p - r * TWH;
                    const int iy = iy0 + r, ix = ix0 + c;
                    dsto[u] = valid ? p * XCP + c2 : -1;
                    v[u] = make_float2(0.0f, 0.0f);
                    mv[u] = make_float2(1.0f, 1.0f);
                    if (valid && chok && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win) {
                        const size_t o = ((size_t)(iy >> up) * Ws + (ix >> up)) * C + chl;
                        v[u] = *(const float2*)(sb + o);
                        if (mb) mv[u] = *(const float2*)(mb + o);
                    }
                }
#pragma unroll
                for (int u = 0; u < SM; ++u) {
                    float2 x = v[u];
                    if (a.in_relu) { x.x = x.x > 0.0f ? x.x : 0.0f; x.y = x.y > 0.0f ? x.y : 0.0f; }
                    x.x = mv[u].x > 0.0f ? x.x : 0.0f;
                    x.y = mv[u].y > 0.0f ? x.y : 0.0f;
                    if (dsto[u] >= 0) *(float2*)(xt + dsto[u]) = x;
                }
            }
        } else {
            const int sp = lane >> 4, c1 = lane & 15;
            const int ch = cb + c1;
            const bool chok = ch < Cin;
            const bool second = ch >= a.C0;
            const float* const sb = second ? a.src1 : a.src0;
            const int C = second ? a.C1 : a.C0, chl = second ? ch - a.C0 : ch, up = second ? a.up1 : a.up0;
            const int Ws = a.Win >> up;
            const float* const mb = second ? nullptr : a.mask;
            const int ngrp = (npx + 3) >> 2;
            for (int g0 = wave; g0 < ngrp; g0 += 4 * SU) {
                float v[SU], mv[SU];
                int dsto[SU];
#pragma unroll
                for (int u = 0; u < SU; ++u) {
                    const int grp = g0 + 4 * u, p = grp * 4 + sp;
                    const bool valid = grp < ngrp && p < npx;
                    const int r = (int)(((unsigned)p * inv_twh) >> 20), c = p - r * TWH;
                    const int iy = iy0 + r, ix = ix0 + c;
                    dsto[u] = valid ? p * XCP + c1 : -1;
                    v[u] = 0.0f;
                    mv[u] = 1.0f;
                    if (valid && chok && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win) {
                        const size_t o = ((size_t)(iy >> up) * Ws + (ix >> up)) * C + chl;
                        v[u] = sb[o];
                        if (mb) mv[u] = mb[o];
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

        // ---- the slab's k-steps: (ky, kx, four channels).  The pipeline unit is a TAP: the (up to) four k-steps of a tap are
        // requested while the MFMAs of the tap before it run (two register sets).  No vector ALU instruction per fragment:
        //   * pixel fragments: ds_read_b32 with the k-step's channel offset as the instruction's immediate; the per-lane
        //     base address moves once per tap (MT adds per 4 x MT x NT MFMAs);
        //   * weight fragments: buffer loads -- per-lane byte offset fixed for the whole kernel (voff), the (tap, slab,
        //     k-step) part in the scalar offset; lanes without a weight (cout padding) carry an out-of-range offset, for
        //     which the hardware returns 0, and the channel padding of the last slab needs no select at all: its x is
        //     the slab's zero fill, and 0 * (a finite neighbouring weight, or the 0 behind the buffer's end) adds nothing.
        // (The first version computed a 64-bit address and an exec-masked select per weight load and a shift-add per LDS
        // read: ~22 vector ALU instructions per 8 MFMAs.  Vector ALU and matrix instructions share a SIMD's issue port:
        // with the fragment loads switched off that build ran 1131 instead of 1430 us on conv2 -- tools/ab_f32.sh.)
        (void)cn;
        const int KHW = a.deconv4 ? 1 : a.KH * a.KW;
        const int KWe = a.deconv4 ? 1 : a.KW;
        int kx_ = 0, toff = 0;                                 // wave-uniform walk state: LDS float offset of the tap ...
        unsigned wso = (unsigned)(cb * a.Cout) * 4u;            // ... and byte offset of (tap, slab) in the weight buffer
        const unsigned wstep = (unsigned)(Cin * a.Cout) * 4u, kstep = (unsigned)(4 * a.Cout) * 4u;
        float xf[4][MT], wf[4][NT];                             // fragments of the tap's (up to) four k-steps
        const int ntap = (PSEG_DIAG && (a.dbg & 4)) ? 0 : KHW;  // dbg & 4: no k-loop
        // NKS = k-steps per tap, a compile-time constant inside the loop (straight-line tap bodies: with a run-time count
        // the per-k-step branches cut the body into basic blocks and the compiler issued each tap's loads right in front
        // of its own MFMAs).  ONE register set, one tap of lead: k-step s of tap T + 1 is requested into the registers of
        // k-step s of tap T as soon as that step's MFMAs have been issued.
        auto run_ksteps = [&](auto nks_c) {
            constexpr int NKS = decltype(nks_c)::value;
            const float* xp[MT];
            auto point = [&]() {                                // fragment pointers / weight offset of the tap the walk is at
#pragma unroll
                for (int m = 0; m < MT; ++m) xp[m] = xt + xbase[m] + toff;
            };
            auto advance = [&]() {
                toff += XCP;
                wso += wstep;
                if (++kx_ == KWe) { kx_ = 0; toff += (TWH - KWe) * XCP; }
            };
            auto request = [&](int s4, unsigned wo) {           // k-step s4 of the tap `xp` / `wo` point at
                if (PSEG_DIAG && (a.dbg & 8)) {                 // timing ablation: no weight loads
#pragma unroll
                    for (int t = 0; t < NT; ++t) wf[s4][t] = (float)(wo + t);
                } else {
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        wf[s4][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wrsrc, voff[t], wo + (unsigned)s4 * kstep, 0));
                }
                if (PSEG_DIAG && (a.dbg & 16)) {                // timing ablation: no LDS fragment reads
#pragma unroll
                    for (int m = 0; m < MT; ++m) xf[s4][m] = (float)(wo + m);
                } else {
#pragma unroll
                    for (int m = 0; m < MT; ++m) xf[s4][m] = xp[m][4 * s4];
                }
            };
            auto mma = [&](int s4) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s4][t], xf[s4][m], acc[m][t], 0, 0, 0);
            };
            if (ntap <= 0) return;
            point();
#pragma unroll
            for (int s4 = 0; s4 < NKS; ++s4) request(s4, wso);
            for (int tap = 0; tap + 1 < ntap; ++tap) {
                advance();
                point();                                         // the NEXT tap
#pragma unroll
                for (int s4 = 0; s4 < NKS; ++s4) {
                    mma(s4);
                    request(s4, wso);
                }
            }
#pragma unroll
            for (int s4 = 0; s4 < NKS; ++s4) mma(s4);
        };
        run_ksteps(nks_tag);

        // ---- the left-over channels' tile(s): the same walk over KH x (KW + DX - 1) taps of the shifted kernel
        if constexpr (REM != 0) {
            constexpr int NKS = decltype(nks_tag)::value;
            int kxr = 0, toffr = 0;
            unsigned wsr = (unsigned)(cb * 16) * 4u;
            const unsigned wstepr = (unsigned)(Cin * 16) * 4u, kstepr = 4u * 16u * 4u;
            const int ntapr = (PSEG_DIAG && (a.dbg & 4)) ? 0 : a.KH * KWR;
            float xq[4][RT], wq[4];
            const float* xpr[RT];
            auto pointr = [&]() {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) xpr[rt] = xt + xrbase[rt] + toffr;
            };
            auto advancer = [&]() {
                toffr += XCP;
                wsr += wstepr;
                if (++kxr == KWR) { kxr = 0; toffr += (TWH - KWR) * XCP; }
            };
            auto requestr = [&](int s4, unsigned wo) {
                wq[s4] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wrrsrc, voffr, wo + (unsigned)s4 * kstepr, 0));
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) xq[s4][rt] = xpr[rt][4 * s4];
            };
            auto mmar = [&](int s4) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) accr[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[s4], xq[s4][rt], accr[rt], 0, 0, 0);
            };
            if (ntapr > 0) {
                pointr();
#pragma unroll
                for (int s4 = 0; s4 < NKS; ++s4) requestr(s4, wsr);
                for (int tap = 0; tap + 1 < ntapr; ++tap) {
                    advancer();
                    pointr();
#pragma unroll
                    for (int s4 = 0; s4 < NKS; ++s4) {
                        mmar(s4);
                        requestr(s4, wsr);
                    }
                }
#pragma unroll
                for (int s4 = 0; s4 < NKS; ++s4) mmar(s4);
            }
        }
    };
    const int nfull = (TAILK == 0) ? (Cin + XCB - 1) / XCB : Cin / XCB;     // (TAILK == 0: a last slab of 13..15 channels also takes four k-steps)
    for (int sb = 0; sb < nfull; ++sb) process_slab(sb * XCB, std::integral_constant<int, 4>{});
    if constexpr (TAILK != 0) process_slab(nfull * XCB, std::integral_constant<int, TAILK>{});

    // ---- epilogue: acc + bias (+ add), ReLU; lane owns n = 4g..4g+3 of pixel p16 in every tile.  The lane's four values are
    // consecutive output channels of one pixel (of one sub-pixel, for the transposed GEMM: Cout % 4 == 0 keeps a quad inside
    // its sub-pixel): one 16-byte (or two 8-byte, Cout even) store instead of four 4-byte ones.
    const int osy = a.out_sy ? a.out_sy : 1, osx = a.out_sx ? a.out_sx : 1;
    const int pitch = a.dst_pitch ? a.dst_pitch : a.Wout;
    const int vecw = (a.Cout & 3) == 0 ? 4 : ((a.Cout & 1) == 0 ? 2 : 1);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int y = oy0 + wave * RW + (m >> 1), x = ox0 + (m & 1) * 16 + p16;
        if (y >= a.Hout || x >= a.Wout || (PSEG_DIAG && (a.dbg & 2) && acc[m][0][0] != 123.456f)) continue;   // dbg & 2: no stores
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
    if constexpr (REM != 0) {
        // left-over channels: lane (p16, g) holds rows 4g .. 4g+3 = channels 16 NT + j0 .. + 3 of pixel DX * column + dx
        const int dx = (4 * g) / REM, j0 = (4 * g) % REM;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int y = REM == 8 ? oy0 + wave * RW + rt : oy0 + wave * RW + rt * 2 + (p16 >> 3);
            const int x = REM == 8 ? ox0 + 2 * p16 + dx : ox0 + 4 * (p16 & 7) + dx;
            if (y >= a.Hout || x >= a.Wout || (PSEG_DIAG && (a.dbg & 2) && accr[rt][0] != 123.456f)) continue;
            const int n0 = NT * 16 + j0;
            const size_t off0 = ((size_t)y * pitch + (size_t)x) * a.Cout + n0;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = a.bias ? accr[rt][r] + a.bias[n0 + r] : accr[rt][r];
            if (a.add) {
                const float4 ad = *(const float4*)(a.add + off0);
                v[0] = v[0] + ad.x; v[1] = v[1] + ad.y; v[2] = v[2] + ad.z; v[3] = v[3] + ad.w;
            }
            if (a.relu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.0f ? v[r] : 0.0f;
            }
            *(float4*)(a.dst + off0) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
    // ---- fused MaxPooling2D 2x2 (lib/model.py:54,59,64) of the tensor just stored: a wave holds both rows of its pixel
    // pairs (accumulator tiles m and m + 2), the horizontal neighbour sits in lane p16 ^ 1.  The same comparisons in the same
    // order as pool_exact_kernel -- (x, x+1) of the upper row, of the lower row, then the two winners -- so the same bits
    // (signed zeros included); the separate pass re-read the whole tensor (377 MB after conv2).
    if constexpr (MT == 4) {
        if (a.pool_dst) {
            const int Wp = a.Wout >> 1;
            const int yp = (oy0 >> 1) + wave;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int x = ox0 + c * 16 + p16;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int n0 = co_base + t * 16 + 4 * g;
                    float o[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int co = n0 + r < Ntot ? n0 + r : 0;
                        float u0 = a.bias ? acc[c][t][r] + a.bias[co] : acc[c][t][r];
                        float u1 = a.bias ? acc[2 + c][t][r] + a.bias[co] : acc[2 + c][t][r];
                        if (a.relu) { u0 = u0 > 0.0f ? u0 : 0.0f; u1 = u1 > 0.0f ? u1 : 0.0f; }
                        const float r0 = __shfl_xor(u0, 1), r1 = __shfl_xor(u1, 1);
                        const float mt = u0 > r0 ? u0 : r0, mb = u1 > r1 ? u1 : r1;
                        o[r] = mt > mb ? mt : mb;
                    }
                    if ((p16 & 1) || 2 * yp + 1 >= a.Hout || x + 1 >= a.Wout) continue;
                    if (n0 >= Ntot) continue;
                    float* od = a.pool_dst + ((size_t)yp * Wp + (x >> 1)) * a.Cout + n0;
                    if (vecw == 4 && n0 + 3 < Ntot) {
                        *(float4*)od = make_float4(o[0], o[1], o[2], o[3]);
                    } else if (vecw >= 2) {
                        if (n0 + 1 < Ntot) *(float2*)od = make_float2(o[0], o[1]);
                        else od[0] = o[0];
                        if (n0 + 3 < Ntot) *(float2*)(od + 2) = make_float2(o[2], o[3]);
                        else if (n0 + 2 < Ntot) od[2] = o[2];
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (n0 + r < Ntot) od[r] = o[r];
                    }
                }
            }
        }
    }
}

template <int MT, int NT, int TAILK, int REM = 0>
static int launch_xb(const ConvArgs& a, int THH, int TWH, dim3 grid, size_t lds, hipStream_t st) {
    static bool attr_set[64] = {false};
    int dev = 0;
    PSEG_HIP(hipGetDevice(&dev));
    if (!attr_set[dev & 63]) {
        PSEG_HIP(hipFuncSetAttribute((const void*)conv_xb_kernel<MT, NT, TAILK, REM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[dev & 63] = true;
    }
    conv_xb_kernel<MT, NT, TAILK, REM><<<grid, 256, lds, st>>>(a, THH, TWH, (1u << 20) / (unsigned)TWH + 1u);
    PSEG_HIP(hipGetLastError());
    return PSEG_OK;
}

// wr[ky][kxp][ci][dx * rem + j] = w[ky][kxp - dx][ci][cmain + j] (0 when kxp - dx is not a tap): the left-over channels'
// kernel as DX shifted copies, one per pixel of the group a tile column stands for (see conv_xb_kernel, REM)
__global__ void wrem_kernel(const float* w, int KH, int KW, int Cin, int Cout, int cmain, int rem, float* wr) {
    const int dxn = 16 / rem, KWR = KW + dxn - 1;
    const int n = KH * KWR * Cin * 16;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int row = i & 15, ci = (i >> 4) % Cin, t = (i >> 4) / Cin;
        const int kxp = t % KWR, ky = t / KWR;
        const int dx = row / rem, j = row - dx * rem, kx = kxp - dx;
        wr[i] = (kx >= 0 && kx < KW) ? w[((size_t)(ky * KW + kx) * Cin + ci) * Cout + cmain + j] : 0.0f;
    }
}

size_t wrem_bytes_for(int KH, int KW, int Cin, int Cout) {
    const int rem = Cout % 16, ntm = Cout / 16;
    if (!((rem == 4 || rem == 8) && (ntm == 1 || ntm == 2))) return 0;
    return (size_t)KH * (KW + 16 / rem - 1) * Cin * 16 * 4;
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

// Returns 1 when the layer was launched on a matrix-core kernel (2: with ConvArgs.pool_dst written too), 0 when it is not
// one for them (caller falls back to the 1x1 / scalar kernels: a handful of output channels), < 0 on error.
int launch_conv_exact_mfma(const ConvArgs& a_in, hipStream_t st) {
    ConvArgs a = a_in;
    if (const char* dv = PSEG_DIAG_KNOB("PSEG_XM_DBG")) a.dbg = atoi(dv);   // wrong-result timing switches: diagnostic build only
    const int Cin = a.C0 + a.C1;
    if (Cin < 1 || a.Cout < 1 || a.KH != a.KW) return 0;
    const int Ntot = a.deconv4 ? 4 * a.Cout : a.Cout;
    if (Ntot < 8) return 0;                    // a handful of couts (logits): the 1x1 / scalar kernels waste less
    const int ntall = cdiv(Ntot, 16);
    const int TWH = (XTW - 1) * a.stride + a.KW;
    if (Cin < 8) {
        // first layers: K flattened across taps (one slab: the oracle's order), all-channel tile
        if (a.deconv4) return 0;
        a.pool_dst = nullptr;                  // (no fused pool on this path: the caller sees return code 1 and pools itself)
        int Cp = std::max(Cin, 2);
        while (Cp % 4 != 2) ++Cp;
        const int THH = (8 - 1) * a.stride + a.KH;
        const size_t lds = (size_t)THH * TWH * Cp * 4;
        if (lds > 150 * 1024) return 0;
        const int NT = ntall <= 4 ? ntall : (ntall == 5 ? 5 : 4);
        dim3 grid(cdiv(a.Wout, XTW) * cdiv(a.Hout, 8), cdiv(ntall, NT));
        switch (NT) {
            case 1: PSEG_TRY((launch_xm<4, 1, true>(a, Cp, THH, TWH, Cin, grid, lds, st))); break;
            case 2: PSEG_TRY((launch_xm<4, 2, true>(a, Cp, THH, TWH, Cin, grid, lds, st))); break;
            case 3: PSEG_TRY((launch_xm<4, 3, true>(a, Cp, THH, TWH, Cin, grid, lds, st))); break;
            case 4: PSEG_TRY((launch_xm<4, 4, true>(a, Cp, THH, TWH, Cin, grid, lds, st))); break;
            default: PSEG_TRY((launch_xm<4, 5, true>(a, Cp, THH, TWH, Cin, grid, lds, st))); break;
        }
        return 1;
    }
    // blocked chain: 8-row tiles unless the 16-channel slab of a strided layer's halo tile would leave fewer than three
    // workgroups per CU (then 4-row tiles)
    auto lds_of = [&](int mt) { return (size_t)((4 * (mt / 2) - 1) * a.stride + a.KH) * TWH * XCP * 4; };
    int MT = (lds_of(4) <= 52 * 1024) ? 4 : 2;
    const bool pooled = a.pool_dst != nullptr && MT == 4 && a.stride == 1 && !a.deconv4 && !a.add && !(a.Hout & 1) && !(a.Wout & 1);
    if (!pooled) a.pool_dst = nullptr;
    if (lds_of(MT) > 150 * 1024) return 0;
    const int TH = 4 * (MT / 2);
    const int THH = (TH - 1) * a.stride + a.KH;
    const int tiles = cdiv(a.Wout, XTW) * cdiv(a.Hout, TH);
    // cout tiles per workgroup: at most four (64 accumulator registers), split evenly over the cout blocks; small layers
    // (1/8-resolution: 192 tiles for 256 CUs) split further until the chip is full -- restaging a 31 KB slab per cout
    // block costs little next to 25 taps of float32 MFMAs
    int nblk = cdiv(ntall, 4), NT = cdiv(ntall, nblk);
    while (NT > 1 && tiles * nblk < 1024) { --NT; nblk = cdiv(ntall, NT); }
    NT = cdiv(ntall, nblk);                    // even split: 4 tiles over 2 blocks are 2 + 2, not 3 + 1 (idle MFMAs on the empty tiles)
    // quarter-resolution layers of a 2048x1536 page (768 tiles): the chip holds 768 three- or four-tile workgroups at once (three
    // waves per SIMD), so one cout block beats two -- 3 tiles cannot split evenly (2 + 1 with a dead tile: deconv3 612 -> 461 us),
    // 4 tiles as 2 + 2 stage every slab twice (conv5 / conv6 226 -> 218, 326 -> 308 us).  PSEG_EXACT_SPLIT=1: the split forms.
    if ((ntall == 3 || ntall == 4) && nblk == 2 && tiles >= 512) { NT = ntall; nblk = 1; }
    const size_t lds = lds_of(MT);
    dim3 grid(tiles, nblk);
    const int tail_ch = Cin % XCB;                              // channels of the last slab (0: a full one)
    const int tailk = (tail_ch == 0 || tail_ch > 12) ? 0 : (tail_ch + 3) / 4;
    // left-over output channels (Cout = 16 k + 4 or + 8) as shifted-pixel tiles instead of a padded cout tile
    {
        const int rem = a.Cout % 16, ntm = a.Cout / 16;
        if ((rem == 4 || rem == 8) && (ntm == 1 || ntm == 2) && MT == 4 && !a.deconv4 && a.stride == 1 && !a.out_sy && !a.out_sx && (tiles >= 512 || PSEG_KNOB("PSEG_EXACT_REM_ANY")) &&
            !PSEG_KNOB("PSEG_EXACT_NO_REM") && a.wrem_buf && a.wrem_cap >= wrem_bytes_for(a.KH, a.KW, Cin, a.Cout)) {
            const size_t wbytes = wrem_bytes_for(a.KH, a.KW, Cin, a.Cout);
            // the shifted copies live in a buffer of the caller's (the op's, beside its kernel; the train step's scratch for the
            // flipped kernels of the data gradients) and are rebuilt only when the weights behind them changed -- no allocation,
            // no pool, nothing but kernels on `st`
            float* const wr = a.wrem_buf;
            if (!a.wrem_valid || !*a.wrem_valid) {
                wrem_kernel<<<(int)std::min<size_t>((wbytes / 4 + 255) / 256, 1024), 256, 0, st>>>(a.w, a.KH, a.KW, Cin, a.Cout, ntm * 16, rem, wr);
                if (a.wrem_valid) *a.wrem_valid = true;
            }
            a.wrem = wr;
            a.pool_dst = nullptr;                               // (the caller pools: return code 1)
            const dim3 g1(tiles, 1);
            int rc = PSEG_OK;
#define PSEG_XR(NT_, REM_)                                                                        \
            if (ntm == NT_ && rem == REM_) {                                                          \
                switch (tailk) {                                                                      \
                    case 0: rc = launch_xb<4, NT_, 0, REM_>(a, THH, TWH, g1, lds, st); break;         \
                    case 1: rc = launch_xb<4, NT_, 1, REM_>(a, THH, TWH, g1, lds, st); break;         \
                    case 2: rc = launch_xb<4, NT_, 2, REM_>(a, THH, TWH, g1, lds, st); break;         \
                    default: rc = launch_xb<4, NT_, 3, REM_>(a, THH, TWH, g1, lds, st); break;        \
                }                                                                                     \
            }
            PSEG_XR(1, 4) PSEG_XR(2, 8) PSEG_XR(1, 8) PSEG_XR(2, 4)
#undef PSEG_XR
            if (rc != PSEG_OK) return rc;
            return 1;
        }
    }
#define PSEG_XB(MT_, NT_)                                                                        \
    if (MT == MT_ && NT == NT_) {                                                                \
        switch (tailk) {                                                                         \
            case 0: PSEG_TRY((launch_xb<MT_, NT_, 0>(a, THH, TWH, grid, lds, st))); break;       \
            case 1: PSEG_TRY((launch_xb<MT_, NT_, 1>(a, THH, TWH, grid, lds, st))); break;       \
            case 2: PSEG_TRY((launch_xb<MT_, NT_, 2>(a, THH, TWH, grid, lds, st))); break;       \
            default: PSEG_TRY((launch_xb<MT_, NT_, 3>(a, THH, TWH, grid, lds, st))); break;      \
        }                                                                                        \
        return pooled ? 2 : 1;                                                                   \
    }
    PSEG_XB(4, 1) PSEG_XB(4, 2) PSEG_XB(4, 3) PSEG_XB(4, 4)
    PSEG_XB(2, 1) PSEG_XB(2, 2) PSEG_XB(2, 3) PSEG_XB(2, 4)
#undef PSEG_XB
    return 0;
}

}  // namespace pseg
