// pseg_exact_valu.hip -- float32-exact layers that are HBM-bound, on the vector ALU.
//
// The first layer (1 or 3 input channels), the k2 s2 transposed convolutions and the logits layer have 1-2 MFLOP per KB of
// tensor traffic: their time is the tensors they stream, not their arithmetic (fcn_skip at 2048x1536: conv1 writes 252 MB
// for 3 GFLOP; deconv5 + logits read 600 MB and write 290 MB for 9.7 GFLOP).  On the matrix-core kernels they ran at
// 0.2-0.7 TB/s: 16 x 16 accumulator tiles scatter 16-byte pieces of 80-byte pixels, every layer re-reads what the one
// before wrote.  Here a thread owns one OUTPUT PIXEL and all of its channels:
//   * the accumulation chain is the oracle's, literally: acc = fmaf(x, w, acc) in (slab of 16 channels, ky, kx, ci) order
//     -- for these layers (one tap, or fewer than 16 input channels) that is plain ascending order -- then + bias, ReLU;
//   * weights are indexed by loop counters and blockIdx only: wave-uniform, served through the scalar cache as the
//     second operand of v_fmac_f32 (a wave of a transposed conv holds pixels of ONE sub-pixel parity);
//   * inputs come through LDS with coalesced loads, outputs leave as whole pixels (CT consecutive floats per thread);
//   * the fused tail computes Conv2DTranspose k2 s2 (deconv5), stores it (the train step and pseg_get_activation read it),
//     and runs the 1x1 logits layer over [deconv5, skip] + argmax on the values still in registers: the 252 MB tensor is
//     not read back, the float32 logits are written only when the caller asks for them.
#include <algorithm>

#include "pseg_common.h"

namespace pseg {

// CT consecutive floats of a wave-uniform row: one batch of wide scalar loads (an index select per element -- "co < Cout ?
// co : 0" -- made them twenty dependent single loads with a wait each)
template <int CT>
__device__ __forceinline__ void ld_row(float (&w)[CT], const float* row) {
#pragma unroll
    for (int j = 0; j < CT; ++j) w[j] = row[j];
}

// ---- first layer: Cin <= 3, k x k, stride 1 ---------------------------------------------------------------------------
constexpr int FV_TH = 8, FV_TW = 32;      // output tile of a 256-thread workgroup: one pixel per thread

template <int CT>
__global__ __launch_bounds__(256) void conv_first_valu_kernel(ConvArgs a) {
    extern __shared__ float xs[];          // [(FV_TH + K - 1)][(FV_TW + K - 1)][Cin]
    const int K = a.KH, Cin = a.C0;
    const int TWH = FV_TW + K - 1, THH = FV_TH + K - 1;
    const int tiles_x = (a.Wout + FV_TW - 1) / FV_TW;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int oy0 = ty * FV_TH, ox0 = tx * FV_TW;
    const int co0 = blockIdx.y * CT;
    // stage the halo tile: rows of TWH * Cin contiguous floats
    const int rowf = TWH * Cin;
    for (int i = threadIdx.x; i < THH * rowf; i += 256) {
        const int r = i / rowf, e = i - r * rowf;
        const int iy = oy0 - a.pt + r, ix = ox0 - a.pl + e / Cin;
        xs[i] = (iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win) ? a.src0[((size_t)iy * a.Win + (ox0 - a.pl)) * Cin + e] : 0.0f;
    }
    __syncthreads();
    const int ly = threadIdx.x >> 5, lx = threadIdx.x & 31;
    const int y = oy0 + ly, x = ox0 + lx;
    float acc[CT];
#pragma unroll
    for (int j = 0; j < CT; ++j) acc[j] = 0.0f;
    const float* wt = a.w + co0;
    for (int ky = 0; ky < K; ++ky)
        for (int kx = 0; kx < K; ++kx) {
            const float* xp = xs + ((ly + ky) * TWH + lx + kx) * Cin;
            for (int ci = 0; ci < Cin; ++ci) {
                const float xv = xp[ci];
                const float* wr = wt + (size_t)((ky * K + kx) * Cin + ci) * a.Cout;
#pragma unroll
                for (int j = 0; j < CT; ++j) acc[j] = __builtin_fmaf(xv, wr[j], acc[j]);     // (out-of-image x = 0: acc unchanged)
            }
        }
    // Whole-pixel outputs without an addend leave through LDS: a lane's CT floats are 80-128 bytes at an 80-128-byte stride, so a
    // direct store instruction touches 64 cache lines for 16 bytes each; transposed through a per-wave LDS patch the same bytes go
    // out as fully contiguous 1 KiB store instructions (a wave's two tile rows are two contiguous runs of 32 pixels).
    const bool lds_store = CT == a.Cout && !a.add && (CT & 1) == 0;
    if (!lds_store && (y >= a.Hout || x >= a.Wout)) return;
    const size_t opix = (size_t)y * (a.dst_pitch ? a.dst_pitch : a.Wout) + x;
    float* o = a.dst + opix * a.Cout + co0;
    float v[CT];
    const float* ad = a.add ? a.add + opix * a.Cout + co0 : nullptr;     // residual addend / data-gradient accumulation (may alias dst)
    const bool whole = co0 + CT <= a.Cout;
    const bool vec4 = (a.Cout & 3) == 0 && (CT & 3) == 0 && whole, vec2 = (a.Cout & 1) == 0 && (CT & 1) == 0 && whole;
    float av[CT];
#pragma unroll
    for (int j = 0; j < CT; ++j) av[j] = 0.0f;
    if (ad) {                                                            // (a pixel's channels are one contiguous run: wide loads)
        if (vec4) {
#pragma unroll
            for (int j = 0; j < CT; j += 4) { const float4 q = *(const float4*)(ad + j); av[j] = q.x; av[j + 1] = q.y; av[j + 2] = q.z; av[j + 3] = q.w; }
        } else if (vec2) {
#pragma unroll
            for (int j = 0; j < CT; j += 2) { const float2 q = *(const float2*)(ad + j); av[j] = q.x; av[j + 1] = q.y; }
        } else {
#pragma unroll
            for (int j = 0; j < CT; ++j)
                if (co0 + j < a.Cout) av[j] = ad[j];
        }
    }
    float bz[CT];
#pragma unroll
    for (int j = 0; j < CT; ++j) bz[j] = 0.0f;
    if (a.bias) {
        if (whole) ld_row<CT>(bz, a.bias + co0);
        else {
#pragma unroll
            for (int j = 0; j < CT; ++j) bz[j] = a.bias[co0 + j < a.Cout ? co0 + j : 0];
        }
    }
#pragma unroll
    for (int j = 0; j < CT; ++j) {
        float t = a.bias ? acc[j] + bz[j] : acc[j];
        if (ad) t = t + av[j];
        if (a.relu) t = t > 0.0f ? t : 0.0f;
        v[j] = t;
    }
    if (lds_store) {
        constexpr int V = (CT & 3) == 0 ? 4 : 2;                  // floats per vector
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        float* ys = xs + ((THH * rowf + 3) & ~3) + wave * (64 * CT);     // this wave's patch: [2 rows x 32 pixels][CT]
        {
            float* d = ys + lane * CT;                             // lane = (row & 1) * 32 + lx: the patch is in memory order
#pragma unroll
            for (int j = 0; j < CT; j += V) {
                if (V == 4) *(float4*)(d + j) = make_float4(v[j], v[j + 1], v[j + 2], v[j + 3]);
                else *(float2*)(d + j) = make_float2(v[j], v[j + 1]);
            }
        }
        const int pitch = a.dst_pitch ? a.dst_pitch : a.Wout;
        constexpr int RV = 32 * CT / V;                            // vectors per tile row
#pragma unroll
        for (int u = 0; u < CT / V; ++u) {
            const int i = lane + 64 * u, rr = i / RV, k = i - rr * RV;
            const int yy = oy0 + 2 * wave + rr, px = (k * V) / CT;
            if (yy < a.Hout && ox0 + px < a.Wout) {
                float* g = a.dst + ((size_t)yy * pitch + ox0) * CT + (size_t)k * V;
                if (V == 4) *(float4*)g = *(const float4*)(ys + i * 4);
                else *(float2*)g = *(const float2*)(ys + i * 2);
            }
        }
    } else if (vec4) {
#pragma unroll
        for (int j = 0; j < CT; j += 4) *(float4*)(o + j) = make_float4(v[j], v[j + 1], v[j + 2], v[j + 3]);
    } else if (vec2) {
#pragma unroll
        for (int j = 0; j < CT; j += 2) *(float2*)(o + j) = make_float2(v[j], v[j + 1]);
    } else {
#pragma unroll
        for (int j = 0; j < CT; ++j)
            if (co0 + j < a.Cout) o[j] = v[j];
    }
}

// ---- Conv2DTranspose k2 s2 (lib/model.py:71,79,83), optionally with the logits layer + argmax behind it -----------------
constexpr int DV_PX = 64;     // input pixels per workgroup: four waves = the four sub-pixel parities of the same 64 pixels

template <int CT, int NCT>      // NCT: 0 = plain transposed conv; else logits classes rounded up to 3 / 4 / 6 / 8
__global__ __launch_bounds__(256) void deconv2_valu_kernel(TailArgs a) {
    constexpr bool TAIL = NCT > 0;
    extern __shared__ float xs[];          // [DV_PX][Cin + 1]; the tail reuses it as [4 * DV_PX][Cs + 1] skip pixels
    const int Cin = a.C0 + a.C1, P = Cin + 1;
    // a workgroup takes (up to) DV_PX consecutive pixels of ONE input row: its outputs are two runs of 2 * np pixels
    const int segs = (a.Win + DV_PX - 1) / DV_PX;
    const int i = blockIdx.x / segs, jseg = (blockIdx.x - i * segs) * DV_PX;
    const int p0 = i * a.Win + jseg;
    const int np = min(DV_PX, a.Win - jseg);
    for (int srcsel = 0; srcsel < (a.C1 > 0 ? 2 : 1); ++srcsel) {
        const int C = srcsel ? a.C1 : a.C0, cbase = srcsel ? a.C0 : 0;
        const float* p = (srcsel ? a.src1 : a.src0) + (size_t)p0 * C;
        const int n = np * C;
        if ((C & 1) == 0) {
            // even channel counts: 8-byte loads, one index split per channel PAIR (half the loads and half the vector ALU work of
            // the 4-byte walk; the two floats of a pair belong to one pixel)
            const int n2 = n >> 1, C2 = C >> 1;
            const unsigned long long inv = (1ull << 32) / (unsigned)C2 + 1ull;   // k / C2 exact while k * C2 < 2^32
            const float2* p2 = (const float2*)p;
            for (int k0 = 0; k0 < n2; k0 += 256 * 8) {
                float2 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int k = k0 + u * 256 + (int)threadIdx.x;
                    v[u] = k < n2 ? p2[k] : make_float2(0.0f, 0.0f);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int k = k0 + u * 256 + (int)threadIdx.x;
                    const int px = (int)(((unsigned long long)(unsigned)k * inv) >> 32), c = 2 * (k - px * C2);
                    if (k < n2) { float* d = xs + px * P + cbase + c; d[0] = v[u].x; d[1] = v[u].y; }
                }
            }
            continue;
        }
        const unsigned long long inv = (1ull << 32) / (unsigned)C + 1ull;    // e / C exact while e * C < 2^32
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
    float* const sks = xs;                   // tail: the skip tensor's pixels of this workgroup, [a][2 * lane + b][Cs + 1] -- in the x tile's place once the transposed conv is done with it (50 -> 32 KB of LDS: five workgroups per CU instead of three)
    // Tail: the logits layer's second source for the 4 * np output pixels is two contiguous runs (output rows 2i, 2i + 1) of
    // 2 * np pixels x Cs floats.  They are REQUESTED here, coalesced, into registers and land in LDS only after the transposed
    // conv below has been computed: their latency hides under its ~700 packed FMAs (read per thread straight from memory, a
    // pixel's 120 bytes were fetched line by line, fifteen times over; staged before the compute phase, the workgroup --
    // three per CU at 50 KB of LDS -- sat waiting for them).
    constexpr int SKR = TAIL ? 16 : 1;       // register slots per row: 2 * DV_PX pixels x Cs floats / 256 threads, Cs <= 32
    float skv[2][SKR];
    if constexpr (TAIL) {
        const int n = 2 * np * a.Cs;
#pragma unroll
        for (int arow = 0; arow < 2; ++arow) {
            const float* p = a.skip + ((size_t)(2 * i + arow) * (2 * a.Win) + 2 * jseg) * a.Cs;
            if ((a.Cs & 1) == 0) {                         // even: channel pairs, 8-byte loads (slot u holds pair u * 256 + thread)
                const float2* p2 = (const float2*)p;
#pragma unroll
                for (int u = 0; u < SKR / 2; ++u) {
                    const int k = u * 256 + (int)threadIdx.x;
                    const float2 q = (a.Cs > 0 && 2 * k < n) ? p2[k] : make_float2(0.0f, 0.0f);
                    skv[arow][2 * u] = q.x; skv[arow][2 * u + 1] = q.y;
                }
                continue;
            }
#pragma unroll
            for (int u = 0; u < SKR; ++u) {
                const int e = u * 256 + (int)threadIdx.x;
                skv[arow][u] = (a.Cs > 0 && e < n) ? p[e] : 0.0f;
            }
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int ab = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave = sub-pixel parity: weights stay wave-uniform (scalar loads)
    const int co0 = TAIL ? 0 : blockIdx.y * CT;
    // no early exits: every lane runs the whole (uniform) instruction stream so that all weight reads stay scalar; lanes
    // past the tensor compute on a clamped pixel and store nothing
    const bool live = lane < np;
    const int j0 = jseg + (live ? lane : 0);
    float acc[CT];
#pragma unroll
    for (int j = 0; j < CT; ++j) acc[j] = 0.0f;
    const float* wt = a.w + (size_t)ab * Cin * a.Cout + co0;
    const float* xp = xs + (live ? lane : 0) * P;
    for (int ci = 0; ci < Cin; ++ci) {
        const float xv = xp[ci];
        const float* wr = wt + (size_t)ci * a.Cout;
#pragma unroll
        for (int j = 0; j < CT; ++j) acc[j] = __builtin_fmaf(xv, wr[j], acc[j]);
    }
    const int oy = 2 * i + (ab >> 1), ox = 2 * j0 + (ab & 1);
    const size_t opix = (size_t)oy * (2 * a.Win) + ox;
    float v[CT];
    {
        float bz[CT];
        if (co0 + CT <= a.Cout) ld_row<CT>(bz, a.bias + co0);
        else {
#pragma unroll
            for (int j = 0; j < CT; ++j) bz[j] = a.bias[co0 + j < a.Cout ? co0 + j : 0];
        }
#pragma unroll
        for (int j = 0; j < CT; ++j) {
            float t = acc[j] + bz[j];
            if (a.relu) t = t > 0.0f ? t : 0.0f;
            v[j] = t;
        }
    }
    // whole-pixel outputs leave through LDS (as in the first-layer kernel): the two waves of a sub-pixel row interleave their
    // pixels into one contiguous run of 2 * np pixels in the x tile's place, then every wave stores a contiguous half of a row
    const bool lds_store = CT == a.Cout && (CT & 1) == 0;
    if (lds_store) {
        constexpr int V = (CT & 3) == 0 ? 4 : 2;
        __syncthreads();                                   // every thread has read its x pixel: the tile's LDS is free
        {
            float* d = xs + ((ab >> 1) * 2 * DV_PX + 2 * lane + (ab & 1)) * CT;
#pragma unroll
            for (int j = 0; j < CT; j += V) {
                if (V == 4) *(float4*)(d + j) = make_float4(v[j], v[j + 1], v[j + 2], v[j + 3]);
                else *(float2*)(d + j) = make_float2(v[j], v[j + 1]);
            }
        }
        __syncthreads();
        const int arow = ab >> 1, half = ab & 1;           // this wave stores pixels [half * DV_PX, + DV_PX) of output row 2i + arow
        const float* ps = xs + (arow * 2 * DV_PX + half * DV_PX) * CT;
        float* g = a.dst + ((size_t)(2 * i + arow) * (2 * a.Win) + 2 * jseg + half * DV_PX) * CT;
        const int nvalid = 2 * np - half * DV_PX;          // pixels of this half that exist
#pragma unroll
        for (int u = 0; u < CT / V; ++u) {
            const int k = lane + 64 * u;                   // vector index inside the half row
            if ((k * V) / CT < nvalid) {
                if (V == 4) *(float4*)(g + (size_t)k * 4) = *(const float4*)(ps + k * 4);
                else *(float2*)(g + (size_t)k * 2) = *(const float2*)(ps + k * 2);
            }
        }
    } else if (live) {
        float* o = a.dst + opix * a.Cout + co0;
        if ((a.Cout & 3) == 0 && (CT & 3) == 0 && co0 + CT <= a.Cout) {
#pragma unroll
            for (int j = 0; j < CT; j += 4) *(float4*)(o + j) = make_float4(v[j], v[j + 1], v[j + 2], v[j + 3]);
        } else if ((a.Cout & 1) == 0 && (CT & 1) == 0 && co0 + CT <= a.Cout) {
#pragma unroll
            for (int j = 0; j < CT; j += 2) *(float2*)(o + j) = make_float2(v[j], v[j + 1]);
        } else {
#pragma unroll
            for (int j = 0; j < CT; ++j)
                if (co0 + j < a.Cout) o[j] = v[j];
        }
    }
    if constexpr (TAIL) {
        // logits = Conv2D 1x1 over Concatenate([deconv5, skip]) (lib/model.py:85-88), crop folded into the extent (:86):
        // chain over the concatenated channel index ascending (one tap: the slabs of 16 are consecutive), + bias.  Classes
        // ncls .. NCT-1 ride along on the neighbouring weights (finite; the weight buffer carries zero slack behind its
        // end) and are never looked at.
        {
            __syncthreads();                             // every thread has read its x pixel: the tile's LDS is free
            const int PS = a.Cs + 1;
            const int n = 2 * np * a.Cs;
            if (a.Cs > 0 && (a.Cs & 1) == 0) {
                const int C2 = a.Cs >> 1;
                const unsigned long long inv2 = (1ull << 32) / (unsigned)C2 + 1ull;
#pragma unroll
                for (int u = 0; u < SKR / 2; ++u) {
                    const int k = u * 256 + (int)threadIdx.x;
                    const int px = (int)(((unsigned long long)(unsigned)k * inv2) >> 32), c = 2 * (k - px * C2);
                    if (2 * k < n) {
#pragma unroll
                        for (int arow = 0; arow < 2; ++arow) {
                            float* d = sks + (arow * 2 * DV_PX + px) * PS + c;
                            d[0] = skv[arow][2 * u]; d[1] = skv[arow][2 * u + 1];
                        }
                    }
                }
            } else {
            const unsigned long long inv = a.Cs > 0 ? (1ull << 32) / (unsigned)a.Cs + 1ull : 0ull;
#pragma unroll
            for (int arow = 0; arow < 2; ++arow)
#pragma unroll
                for (int u = 0; u < SKR; ++u) {
                    const int e = u * 256 + (int)threadIdx.x;
                    const int px = (int)(((unsigned long long)(unsigned)e * inv) >> 32), c = e - px * a.Cs;
                    if (e < n) sks[(arow * 2 * DV_PX + px) * PS + c] = skv[arow][u];
                }
            }
            __syncthreads();
        }
        const bool inside = live && oy < a.H && ox < a.W;
        float z[NCT];
#pragma unroll
        for (int c = 0; c < NCT; ++c) z[c] = 0.0f;
        const float* sp = sks + ((ab >> 1) * 2 * DV_PX + 2 * (live ? lane : 0) + (ab & 1)) * (a.Cs + 1);
        if (a.ncls == NCT) {
            // rows of exactly NCT weights are one contiguous block: the deconv channels' CT rows in one batch of wide scalar
            // loads, the skip channels five rows per batch (a row per iteration was a scalar-cache round trip per channel)
            float wl0[CT * NCT];
            ld_row<CT * NCT>(wl0, a.wl);
#pragma unroll
            for (int j = 0; j < CT; ++j)
#pragma unroll
                for (int c = 0; c < NCT; ++c) z[c] = __builtin_fmaf(v[j], wl0[j * NCT + c], z[c]);
            constexpr int G = 5;
            const float* ws = a.wl + (size_t)CT * NCT;
            int s0 = 0;
            for (; s0 + G <= a.Cs; s0 += G) {
                float wg[G * NCT], sv[G];
                ld_row<G * NCT>(wg, ws + (size_t)s0 * NCT);
#pragma unroll
                for (int u = 0; u < G; ++u) sv[u] = sp[s0 + u];
#pragma unroll
                for (int u = 0; u < G; ++u)
#pragma unroll
                    for (int c = 0; c < NCT; ++c) z[c] = __builtin_fmaf(sv[u], wg[u * NCT + c], z[c]);
            }
            for (; s0 < a.Cs; ++s0) {
                const float sv = sp[s0];
#pragma unroll
                for (int c = 0; c < NCT; ++c) z[c] = __builtin_fmaf(sv, ws[(size_t)s0 * NCT + c], z[c]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < CT; ++j) {
                const float* wr = a.wl + (size_t)j * a.ncls;
#pragma unroll
                for (int c = 0; c < NCT; ++c) z[c] = __builtin_fmaf(v[j], wr[c], z[c]);
            }
            for (int s0 = 0; s0 < a.Cs; ++s0) {
                const float sv = sp[s0];
                const float* wr = a.wl + (size_t)(CT + s0) * a.ncls;
#pragma unroll
                for (int c = 0; c < NCT; ++c) z[c] = __builtin_fmaf(sv, wr[c], z[c]);
            }
        }
        int best = 0;
        float bv = z[0] + a.bl[0];
        z[0] = bv;
#pragma unroll
        for (int c = 1; c < NCT; ++c) {
            z[c] = z[c] + a.bl[c < a.ncls ? c : 0];
            if (c < a.ncls && z[c] > bv) { bv = z[c]; best = c; }       // np.argmax: first maximum wins
        }
        if (inside) {
            const size_t q = (size_t)oy * a.W + ox;
            if (a.logits) {
#pragma unroll
                for (int c = 0; c < NCT; ++c)
                    if (c < a.ncls) a.logits[q * a.ncls + c] = z[c];
            }
            if (a.labels) a.labels[q] = best;
            if (a.labels_u8) a.labels_u8[q] = (uint8_t)best;
        }
    }
}

template <typename K>
static int set_lds_attr(K kernel) {
    PSEG_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return PSEG_OK;
}

// 1 = launched, 0 = not a layer for this kernel
int launch_conv_first_valu(const ConvArgs& a, hipStream_t st) {
    if (PSEG_KNOB("PSEG_EXACT_NO_VALU")) return 0;
    if (a.C1 || a.C0 > 3 || a.KH != a.KW || a.stride != 1 || a.up0 || a.in_relu || a.mask || a.deconv4 || a.pool_dst ||
        a.out_sy || a.out_sx || a.Cout < 8 || a.Cout > 256)
        return 0;
    const int K = a.KH;
    size_t lds = (size_t)(FV_TH + K - 1) * (FV_TW + K - 1) * a.C0 * 4;
    int CT = (a.Cout == 20 || a.Cout == 16 || a.Cout == 32) ? a.Cout : ((a.Cout & 31) == 0 ? 32 : ((a.Cout % 20) == 0 ? 20 : ((a.Cout % 30) == 0 ? 30 : 16)));
    dim3 grid(cdiv(a.Wout, FV_TW) * cdiv(a.Hout, FV_TH), cdiv(a.Cout, CT));
    if (CT == a.Cout && !a.add && (CT & 1) == 0) lds = ((lds + 15) & ~(size_t)15) + (size_t)256 * CT * 4;   // the store patches (see the kernel)
    switch (CT) {
        case 16: conv_first_valu_kernel<16><<<grid, 256, lds, st>>>(a); break;
        case 20: conv_first_valu_kernel<20><<<grid, 256, lds, st>>>(a); break;
        case 30: conv_first_valu_kernel<30><<<grid, 256, lds, st>>>(a); break;
        default: conv_first_valu_kernel<32><<<grid, 256, lds, st>>>(a); break;
    }
    PSEG_HIP(hipGetLastError());
    return 1;
}

// Conv2DTranspose k2 s2; with `tail` the logits layer and argmax run behind it (tail.skip etc. filled in).  1 = launched.
int launch_deconv2_valu(const TailArgs& a, bool tail, hipStream_t st) {
    if (PSEG_KNOB("PSEG_EXACT_NO_VALU")) return 0;
    const int Cin = a.C0 + a.C1;
    if (Cin < 1 || Cin > 512 || a.Cout < 4 || (size_t)a.Hin * a.Win > 0x3fffffff) return 0;
    size_t lds = std::max((size_t)DV_PX * (Cin + 1) * 4, tail ? (size_t)4 * DV_PX * (a.Cs + 1) * 4 : (size_t)0);
    {   // the store patch of a whole-pixel instance (CT == Cout: 2 rows x 2 * DV_PX pixels x Cout)
        const int ct = tail ? 20 : ((a.Cout % 30) == 0 ? 30 : ((a.Cout % 20) == 0 ? 20 : ((a.Cout & 31) == 0 ? 32 : 16)));
        if (ct == a.Cout) lds = std::max(lds, (size_t)4 * DV_PX * a.Cout * 4);
    }
    static bool attr[64] = {false};
    int dev = 0;
    PSEG_HIP(hipGetDevice(&dev));
    if (!attr[dev & 63]) {
        PSEG_TRY(set_lds_attr(deconv2_valu_kernel<20, 3>));
        PSEG_TRY(set_lds_attr(deconv2_valu_kernel<20, 4>));
        PSEG_TRY(set_lds_attr(deconv2_valu_kernel<20, 6>));
        PSEG_TRY(set_lds_attr(deconv2_valu_kernel<20, 8>));
        PSEG_TRY(set_lds_attr(deconv2_valu_kernel<20, 0>));
        PSEG_TRY(set_lds_attr(deconv2_valu_kernel<30, 0>));
        PSEG_TRY(set_lds_attr(deconv2_valu_kernel<32, 0>));
        PSEG_TRY(set_lds_attr(deconv2_valu_kernel<16, 0>));
        attr[dev & 63] = true;
    }
    const int nwg = cdiv(a.Win, DV_PX) * a.Hin;
    if (tail) {
        if (a.Cout != 20 || a.ncls < 1 || a.ncls > 8 || a.relu || (a.Cs > 0 && !a.skip) || a.Cs > 32 || lds > 150 * 1024) return 0;
        if (a.ncls <= 3) deconv2_valu_kernel<20, 3><<<dim3(nwg), 256, lds, st>>>(a);
        else if (a.ncls == 4) deconv2_valu_kernel<20, 4><<<dim3(nwg), 256, lds, st>>>(a);
        else if (a.ncls <= 6) deconv2_valu_kernel<20, 6><<<dim3(nwg), 256, lds, st>>>(a);
        else deconv2_valu_kernel<20, 8><<<dim3(nwg), 256, lds, st>>>(a);
    } else {
        const int CT = (a.Cout % 30) == 0 ? 30 : ((a.Cout % 20) == 0 ? 20 : ((a.Cout & 31) == 0 ? 32 : 16));
        const dim3 grid(nwg, cdiv(a.Cout, CT));
        switch (CT) {
            case 30: deconv2_valu_kernel<30, 0><<<grid, 256, lds, st>>>(a); break;
            case 20: deconv2_valu_kernel<20, 0><<<grid, 256, lds, st>>>(a); break;
            case 32: deconv2_valu_kernel<32, 0><<<grid, 256, lds, st>>>(a); break;
            default: deconv2_valu_kernel<16, 0><<<grid, 256, lds, st>>>(a); break;
        }
    }
    PSEG_HIP(hipGetLastError());
    return 1;
}

}  // namespace pseg
