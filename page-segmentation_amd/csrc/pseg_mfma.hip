// pseg_mfma.hip -- bf16 throughput mode: NHWC bf16 activations (channels padded to 8), bf16
// kernels, float32 MFMA accumulation (v_mfma_f32_16x16x32_bf16), gfx950 only.
//
// conv_mfma_kernel: implicit GEMM.  D[cout][pixel] += W[cout][k] * X[k][pixel] with
//   k = (tap, channel-chunk of 8) in a host-chosen order.  Per workgroup (256 threads = 4 waves):
//   * the input halo tile of one channel block is staged once into LDS as [row][pixel][chunk];
//     the pixel stride is sigma*16 B with sigma = 2 (mod 4) and the row pitch an odd number of
//     16-B slots, which makes every ds_read_b128 im2col fragment read bank-conflict-free when
//     the two k-chunks a lane-group pair reads differ by an odd slot count (tools/lds_conflicts.py);
//     the host orders the chunks to satisfy that (pair_chunks()).
//   * weights are pre-packed on the host in MFMA A-fragment order and streamed through LDS in
//     groups of GK k-steps by LDS-DMA (global_load_lds_dwordx4), double-buffered.
//   * each wave owns MT pixel tiles (16 consecutive x) x NT cout tiles (16) of accumulators.
//   * epilogue: +bias, (+residual), ReLU, bf16, transpose through LDS, coalesced 16-B NHWC
//     stores, optional fused 2x2 max-pool output.
//   The same kernel runs Conv2DTranspose k2 s2 as a 1x1 GEMM with N = 4*Cout and a scatter
//   epilogue.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "pseg_common.h"

namespace pseg {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((__vector_size__(2 * sizeof(unsigned int)))) unsigned int u32x2;
typedef __attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int u32x4;

constexpr int TW = 32;          // output tile width (two 16-pixel MFMA column tiles)
constexpr int MAX_TAB = 1024;   // k-chunks per channel block (KS*KS*nc + padding)

// ---- host bf16 helpers -----------------------------------------------------------------------
static inline uint16_t f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u = u + 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float bf2f(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

__device__ __forceinline__ float d_bf2f(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
// two floats -> packed bf16 pair in ONE v_cvt_pk_bf16_f32 (round-to-nearest-even, lo in bits 0..15)
__device__ __forceinline__ uint32_t pk_bf16(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ uint16_t d_f2bf(float f) {
    __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: round-to-nearest-even
    return __builtin_bit_cast(uint16_t, b);
}

// ---------------------------------------------------------------------------------------------
// generic MFMA conv
// ---------------------------------------------------------------------------------------------
struct MConv {
    const uint16_t* src0;
    const uint16_t* src1;
    int nch0, nch1;      // 16-byte chunks per pixel of src0 / src1 (Cs / 8)
    unsigned bytes0, bytes1;  // sizes of the source tensors (buffer descriptors)
    int sigma;           // LDS pixel stride in 16-byte slots (PS2 / 16)
    int up0, up1;
    int Hin, Win, Hout, Wout;
    int stride, pt, pl, in_relu, relu;
    // K structure: nblk channel blocks of nc_full chunks (last: nc_last)
    int nblk, nc_full, nc_last, ks_full, ks_last;
    const int* tab_full;  // [ks_full*4] byte offsets of the k-chunks in the LDS input tile
    const int* tab_last;
    const uint16_t* wpk;  // [kstep][NTtot][64 lanes][8] bf16
    int NTtot;
    const float* bias;    // [NTtot*16]
    // LDS geometry
    int PS2, row_pitch, THH, TWH, GK, NB, G, lds_w_off, lds_tab_off;
    // output
    uint16_t* dst;
    // FL_DQ: Conv2DTranspose k2 s2 fused behind this conv (its accumulators are the deconv's B operands)
    const uint16_t* dq_w; const float* dq_bias; uint16_t* dq_dst; unsigned dq_bytes; int dq_nch, dq_relu;
    uint16_t* dst2;       // second copy of the output stored as max(x, 0) (res_unet: read by the next block's pre-activation conv), or null
    int nch_out;
    uint16_t* pool_dst;
    const uint16_t* add;
    unsigned dst_bytes, pool_bytes;
    int deconv, CoP;
    int nb_loop, nb_total;
    int ntiles;          // FL_PERSIST: tiles of the layer (a workgroup walks tile blockIdx.x, + gridDim.x, ...)
    // fused first layer (FL_FUSE1): conv1 is recomputed on the halo tile from the uint8 page
    const uint8_t* f1_img; int f1_H, f1_W; const uint16_t* f1_wpk; const float* f1_bias; int f1_relu, lds_f1_off;
    const uint16_t* f1_wpk32;  // first-layer kernel as 32x32x16 A fragments [k-step 3][lane][8] (conv12_ws_kernel), or null   // N blocks walked inside one workgroup / N blocks of the layer
    // fused tail (deconv5 -> crop -> logits 1x1 -> softmax/argmax), see tail_epilogue()
    int tail, tail_C, H0, W0, nch_skip;
    const uint16_t* skip;      // full-resolution skip tensor (conv2), or null
    const uint16_t* tail_wa;   // [64][8] bf16: logits weights for the deconv channels, fragment order
    const uint16_t* tail_wb;   // [64][8] bf16: logits weights for the skip channels
    const float* tail_bias;    // [16] logits bias (zero padded)
    float* out_logits; float* out_probs; int64_t* out_labels; uint8_t* out_labels_u8;
    int xq, xr;          // XCD-aware tile order: tiles / 8 and tiles % 8 (xq < 0: off)
    float* skip_logits; int skip_CP;   // FL_SKIPLOG: [pixel][skip_CP] f32 logits contribution of this layer's output (it is not stored itself)
    unsigned long long* trace;  // PSEG_TRACE: per-workgroup s_memtime stamps (diagnostic builds only)
    int dbg;   // ablation bits (PSEG_DBG): 1 skip input staging, 2 skip MFMAs, 4 skip epilogue, 8 skip weight DMA
};

// single v_max_f32: fmaxf() makes hipcc canonicalise both operands first (3 instructions per max
// on MFMA outputs); NaNs are not a concern for the ReLU / max-pool epilogue.
__device__ __forceinline__ float vmax(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// max(v, value of lane ^ 1) in one instruction: DPP quad_perm [1,0,3,2] on the first operand
__device__ __forceinline__ float vmax_xor1(float v) {
    float r;
    asm("v_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v));
    return r;
}
// value of the neighbouring lane (lane ^ 1) by DPP quad_perm [1,0,3,2]: one VALU op, no LDS crossbar
__device__ __forceinline__ float lane_xor1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
}

// Top-2 of a pixel's logits.  A lane holds four of them (classes c0 + r, r = 0..3; classes >= C do not count) and the
// other classes of the pixel sit in the lanes lane ^ 16 / lane ^ 32 when NG > 1 lane groups share a pixel.  bv / bi =
// maximum and its class (first maximum wins, as np.argmax), sv = the runner-up's value (== bv on an exact tie): the
// margin bv - sv is what the label-exact mode thresholds (lib/network.py:259; DESIGN.md "label-exact mode").
template <int NG>
__device__ __forceinline__ void top2_classes(const f32x4 z, int c0, int C, float& bv, int& bi, float& sv) {
    bv = -3.4e38f; sv = -3.4e38f; bi = 0x7fffffff;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const bool ok = c0 + r < C;
        const float v = z[r];
        const bool take = ok & (v > bv);
        sv = take ? bv : ((ok & (v > sv)) ? v : sv);
        bv = take ? v : bv;
        bi = take ? c0 + r : bi;
    }
    if constexpr (NG > 1) {
#pragma unroll
        for (int sh = 16; sh < 16 * NG; sh <<= 1) {
            const float ov = __shfl_xor(bv, sh), os = __shfl_xor(sv, sh);
            const int oi = __shfl_xor(bi, sh);
            const bool take = (ov > bv) | ((ov == bv) & (oi < bi));
            const float lose = take ? bv : ov;
            sv = fmaxf(fmaxf(sv, os), lose);
            bv = take ? ov : bv;
            bi = take ? oi : bi;
        }
    }
}

typedef short pk_i16x2 __attribute__((ext_vector_type(2)));
// ReLU of two packed bf16 values in one v_pk_max_i16: a negative bf16 is a negative int16 (floor = 0), floor = -32768 leaves
// the values alone.  Equals rounding max(v, 0): rounding is monotone and -0.0 (0x8000) becomes +0 either way.
__device__ __forceinline__ uint32_t relu_pk_bf16(uint32_t v, uint32_t floor2) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(pk_i16x2, v), __builtin_bit_cast(pk_i16x2, floor2)));
}

__device__ __forceinline__ uint4 relu_bf16x8(uint4 v) {
    auto f = [](uint32_t w) -> uint32_t {
        if (w & 0x8000u) w &= 0xffff0000u;
        if (w & 0x80000000u) w &= 0x0000ffffu;
        return w;
    };
    return make_uint4(f(v.x), f(v.y), f(v.z), f(v.w));
}

__device__ __forceinline__ uint4 max_bf16x8(uint4 a, uint4 b) {
    auto m = [](uint32_t x, uint32_t y) -> uint32_t {
        const float xl = __uint_as_float(x << 16), yl = __uint_as_float(y << 16);
        const float xh = __uint_as_float(x & 0xffff0000u), yh = __uint_as_float(y & 0xffff0000u);
        const uint32_t lo = (xl > yl ? x : y) & 0xffffu;
        const uint32_t hi = (xh > yh ? x : y) & 0xffff0000u;
        return hi | lo;
    };
    return make_uint4(m(a.x, b.x), m(a.y, b.y), m(a.z, b.z), m(a.w, b.w));
}

// counted wait: at most N vector-memory operations of this wave still outstanding
__device__ __forceinline__ void wait_vmcnt_le(int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// workgroup barrier that does NOT drain the LDS-DMA queue (a plain __syncthreads() emits
// vmcnt(0)): own LDS writes/reads retired, then s_barrier.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

constexpr int STAGE_SLOTS = 12;  // 16-byte loads a lane keeps in flight while staging a tile

// Specialisation: KS_ / ST_ / SG_ / MODE_ / FL_ >= 0 are compile-time constants of the hot layer
// shapes (kernel size, stride, LDS pixel stride in slots, epilogue mode, feature flags); -1 means
// "read it from the argument block" (generic fallback).  With ~50 runtime geometry fields the
// compiler hoists and spills scalars by the hundred; with constants the prologue collapses.
enum { MODE_CONV = 0, MODE_DECONV = 1, MODE_TAIL = 2 };
enum { FL_POOL = 1, FL_ADD = 2, FL_INRELU = 4, FL_UP0 = 8, FL_UP1 = 16, FL_FUSE1 = 32, FL_PERSIST = 64, FL_LOGITS = 128, FL_SKIPLOG = 256, FL_DQ = 512 };
#ifndef PSEG_DIAG
#define PSEG_DIAG 0   // 1: compile the in-kernel trace stamps / ablation switches (diagnostic builds)
#endif

// NW_ = waves per workgroup: 4 (two workgroups per CU) or 8 (one workgroup per CU owning all 160 KiB
// of LDS: a 16-row tile shares one weight stream among twice the pixels -- the mid layers are
// bound by LDS-DMA bytes per CU).
template <int MT, int NT, int KS_, int ST_, int SG_, int MODE_, int FL_, int NW_ = 4>
__global__ __launch_bounds__(NW_ * 64) __attribute__((amdgpu_waves_per_eu((MT == 4 && NT == 4 && NW_ == 4 && (KS_ == 5 || KS_ == 3 || KS_ == 2) && (SG_ == 4 || SG_ == 5) && MODE_ == 0) ? 3 : 2))) void conv_mfma_kernel(MConv a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NW = NW_, NTHR = NW_ * 64;
    constexpr int TH = NW * (MT / 2);  // NW waves x (MT/2) rows
    if (gridDim.z > 1) {
        // a launch over several page slots (pseg_predict_batch): blockIdx.z = the slot; only plain layers are launched this way
        // (sources, output and fused pool -- mfma_op_batchable)
        a.src0 = (const uint16_t*)((const char*)a.src0 + (size_t)blockIdx.z * a.bytes0);
        if (a.src1) a.src1 = (const uint16_t*)((const char*)a.src1 + (size_t)blockIdx.z * a.bytes1);
        a.dst = (uint16_t*)((char*)a.dst + (size_t)blockIdx.z * a.dst_bytes);
        if (a.pool_dst) a.pool_dst = (uint16_t*)((char*)a.pool_dst + (size_t)blockIdx.z * a.pool_bytes);
    }
    constexpr bool FIXED = KS_ > 0;
    const int c_stride = FIXED ? ST_ : a.stride;
    const int c_sigma = FIXED ? SG_ : a.sigma;
    const int c_PS2 = c_sigma * 16;
    const int c_THH = FIXED ? (TH - 1) * ST_ + KS_ : a.THH;
    const int c_TWH = FIXED ? (TW - 1) * ST_ + KS_ : a.TWH;
    const int c_up0 = FIXED ? ((FL_ & FL_UP0) ? 1 : 0) : a.up0;
    const int c_up1 = FIXED ? ((FL_ & FL_UP1) ? 1 : 0) : a.up1;
    const bool c_inrelu = FIXED ? (FL_ & FL_INRELU) != 0 : a.in_relu != 0;
    const bool c_pool = FIXED ? (FL_ & FL_POOL) != 0 : a.pool_dst != nullptr;
    const bool c_add = FIXED ? (FL_ & FL_ADD) != 0 : a.add != nullptr;
    const bool c_deconv = FIXED ? MODE_ != MODE_CONV : a.deconv != 0;
    const bool c_tail = FIXED ? MODE_ == MODE_TAIL : a.tail != 0;
    constexpr bool c_fuse1 = FIXED && (FL_ & FL_FUSE1) != 0;
    // persistent workgroups (single channel block, all weights resident in LDS: NB == 1): the
    // weights and the k-chunk table are staged once, then the workgroup walks several tiles
    constexpr bool c_persist = FIXED && (FL_ & FL_PERSIST) != 0;
    const int c_dbg = PSEG_DIAG ? a.dbg : 0;
    const bool getenv_vgpr_inrelu = (a.dbg & 16) != 0;   // host knob PSEG_INRELU_VGPR: old VGPR staging path for pre-activation ReLU
    // pre-activation ReLU (res_unet) of the compiled instances: on the pixel FRAGMENTS, four v_pk_max_i16 per fragment in the shadow
    // of the k-step's MFMAs (bf16 as int16: negative <=> sign bit), not a pass over the LDS tile behind every channel block's DMA
    // (read-modify-write of the whole tile + a barrier: 637 vs 544 us for the two full-resolution layers on the same input)
    constexpr bool c_relu_frag = FIXED && (FL_ & FL_INRELU) != 0;
    unsigned long long* const c_trace = PSEG_DIAG ? a.trace : nullptr;
    char* in_t = smem;
    char* w_t = smem + a.lds_w_off;
    int* tab_l = (int*)(smem + a.lds_tab_off);

    int tid = threadIdx.x, lane = tid & 63;
    int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: SALU address math
    int p16 = lane & 15, g = lane >> 4;
    const int tiles_x = (a.Wout + TW - 1) / TW;
    int oy0, ox0, iy0, ix0;
    // XCD-aware tile order: the dispatcher deals consecutive workgroups round-robin to the 8 XCDs, each with its own
    // L2.  Workgroup b therefore takes tile number (b / 8) of band (b % 8), bands being contiguous runs of tiles (a
    // bijection on [0, tiles) for any tile count): the tiles whose halos overlap sit in one XCD's L2 instead of being
    // fetched from HBM by several.  xq < 0: plain order.
    auto xcd_tile = [&](int t) {
        if (a.xq < 0) return t;
        const int x = t & 7, j = t >> 3;
        return x * a.xq + min(x, a.xr) + j;
    };
    auto set_tile = [&](int t) {
        t = xcd_tile(t);
        const int ty = t / tiles_x, tx = t - ty * tiles_x;
        oy0 = ty * TH; ox0 = tx * TW;
        iy0 = oy0 * c_stride - a.pt; ix0 = ox0 * c_stride - a.pl;
    };
    set_tile(blockIdx.x);
    const int WBUF = a.GK * NT * 1024;

#define PSEG_STAMP(i) if (c_trace && tid == 0) c_trace[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 12 + (i)] = __builtin_amdgcn_s_memtime();
    PSEG_STAMP(0)

    int pixbase[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int row = wave * (MT / 2) + (m >> 1), col = (m & 1) * 16 + p16;
        // stride 2: the tile is staged with its columns de-interleaved by parity (see the staging below), so consecutive
        // lanes read consecutive pixel slots under every tap, as in a stride-1 layer
        pixbase[m] = row * c_stride * a.row_pitch + col * (c_stride == 2 ? 1 : c_stride) * c_PS2;
    }
    const int W0 = a.Win >> c_up0, W1 = a.Win >> c_up1;

    // ---- weight ring: group q (GK k-steps x NT tiles, zero-padded to full groups on the host)
    // goes to ring slot q % NB by LDS-DMA; every wave issues exactly L = GK*NT/4 loads per group,
    // which is what the counted vmcnt waits below rely on.
    const int L = (a.GK * NT - wave + NW - 1) / NW;   // pieces wave, wave + NW, ... of a group (wave-uniform)
    const int G = a.G;
    // packed weights are laid out [group][N block][piece][lane][8]: one group of one N block is
    // a contiguous run of GK*NT KiB in exactly the order of its LDS ring slot, so the DMA source
    // and destination of piece (wave + 4j) are base + j * 4 KiB.
    // N blocks: normally one per workgroup (blockIdx.y); the fused tail walks all of its N blocks in
    // one workgroup (nb_loop > 1, single channel block) so the input tile is staged once.
    // (also deconv4 of fcn_skip, sigma 14: its two N blocks re-read a 41 MB tensor that no longer sits in L2 by the time the second runs)
    constexpr int NBL = (FIXED && (MODE_ == MODE_TAIL || (MODE_ == MODE_DECONV && SG_ == 14))) ? 2 : 1;   // compile-time: other instances keep a flat body
#pragma unroll 1
    for (int nbi = 0; nbi < NBL; ++nbi) {
    const int nb = blockIdx.y * NBL + nbi;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the bias is the accumulators' start value (the first MFMA's C operand) in every instance, as in conv12_ws_kernel:
    // no epilogue adds it, and all paths of the engine round in one order
    constexpr bool c_bias0 = true;

    const uint16_t* wsrc = a.wpk + ((size_t)nb * a.GK * NT + wave) * 512 + lane * 8;
    const size_t wgstride = (size_t)a.nb_total * a.GK * NT * 512;   // elements per group
    auto stage_w = [&](int q, int slot) {
        if (c_dbg & 8) return;
        const uint16_t* src = wsrc + (size_t)q * wgstride;
        char* dstb = w_t + slot * WBUF + wave * 1024;
        for (int j = 0; j < L; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)j * (NW * 512)),
                                             (__attribute__((address_space(3))) void*)(dstb + j * (NW * 1024)), 16, 0, 0);
    };
    // fused tail: the skip-tensor fragments of this lane are requested now, before stage 1; the
    // barriers of the k-loop keep them above it, so their latency hides under the deconv GEMM.
    constexpr int NABT = (NT == 4 || NT == 8) ? NT / 2 : 1;
    uint4 skf[MT][NABT];
    if constexpr (NT == 4 || NT == 8) {
        if (c_tail) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int hy = oy0 + wave * (MT / 2) + (m >> 1), hx = ox0 + (m & 1) * 16 + p16;
#pragma unroll
                for (int abl = 0; abl < NABT; ++abl) {
                    const int ab = nb * NABT + abl;
                    skf[m][abl] = make_uint4(0, 0, 0, 0);
                    // canvas coordinates are always inside the skip tensor (Hp x Wp); chunk g < nch_skip
                    if (a.skip && g < a.nch_skip && hy < a.Hout && hx < a.Wout)
                        skf[m][abl] = *(const uint4*)(a.skip + ((size_t)(2 * hy + (ab >> 1)) * (2 * a.Wout) + 2 * hx + (ab & 1)) * (a.nch_skip * 8) + g * 8);
                }
            }
        }
    }
    // biases of this lane's couts: requested now, consumed in the epilogue
    float4 biasr[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) biasr[t] = *(const float4*)(a.bias + (nb * NT + t) * 16 + 4 * g);
    if constexpr (c_bias0) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{biasr[n].x, biasr[n].y, biasr[n].z, biasr[n].w};
    }
    const int D = a.NB > 1 ? a.NB - 1 : 1;  // prefetch distance in groups (NB == 1: single resident group)
    for (int q = 0; q < D && q < G; ++q) stage_w(q, q);
    int slot_c = 0;             // ring slot of the group being computed
    int slot_n = D % a.NB;      // ring slot the next prefetch goes to
    PSEG_STAMP(1)

    constexpr int F1_NU = 6;
    float ubf[F1_NU];   // fused first layer: this lane's uint8 page values of the current / next tile
    for (int tile = blockIdx.x;;) {   // one trip unless FL_PERSIST
    const bool first_tile = !c_persist || tile == (int)blockIdx.x;
    if constexpr (c_persist) {
        // make the lane / wave ids opaque per trip: everything derived from them is then recomputed inside
        // the trip (a few VALU ops) instead of being hoisted and kept live across the whole tile loop,
        // which had cost the persistent variant its second wave per SIMD
        asm volatile("" : "+v"(tid), "+v"(lane), "+v"(p16), "+v"(g));
#if !PSEG_DIAG   // (the trace build's extra uses of `wave` make this constraint unsatisfiable for the backend)
        asm volatile("" : "+s"(wave));
#endif
    }
    if (c_persist && !first_tile) {
        set_tile(tile);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
                acc[m][n] = c_bias0 ? f32x4{biasr[n].x, biasr[n].y, biasr[n].z, biasr[n].w} : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    int gq = 0;  // global group index
    long long acc_wait = 0, acc_issue = 0;   // PSEG_DIAG: cycles of wave 0 in group waits / weight-DMA issue
    for (int b = 0; b < a.nblk; ++b) {
        const bool last = (b == a.nblk - 1);
        const int nc = last ? a.nc_last : a.nc_full;
        const int ks = last ? a.ks_last : a.ks_full;
        const int c0 = b * a.nc_full;
        if (b > 0) lds_barrier();  // every wave is done reading the previous block's tile / table
        // k-chunk offset table of this block: fetched now, written to LDS after the tile DMAs
        // have been issued (the fetch latency hides under the DMA issue)
        constexpr int TU = (MAX_TAB + NTHR - 1) / NTHR;
        int tabv[TU];
        {
            const int* tg = last ? a.tab_last : a.tab_full;
            if (first_tile)
#pragma unroll
                for (int u = 0; u < TU; ++u) { const int i = tid + u * NTHR; tabv[u] = i < ks * 4 ? tg[i] : 0; }
        }
        if (b == 0) { PSEG_STAMP(2) }
        // ---- stage the input halo tile of this channel block (zero outside the image) -------
        // A wave owns tile rows wave, wave+4, ...; all of a wave's 16-byte loads (up to
        // STAGE_SLOTS per lane) are issued before the first LDS write so that one memory
        // latency is exposed per chunk instead of one per row.
        if constexpr (c_fuse1) {
            // ---- fused first layer: conv1 (Cin = 1, 5x5, 20 couts) recomputed on the halo tile ---
            // The uint8 page tile (halo + 2) is converted to bf16(x/255) and kept in LDS twice,
            // once shifted by one pixel, so that the 8 consecutive pixels a lane needs for kernel
            // row ky start on a 4-byte boundary in one of the two copies.  K = 5 rows x 8 columns
            // (as conv1_mfma_kernel); the 16x16 result tile has the pixel on the lane and 4 couts
            // in registers = 8 bytes of the [row][pixel][chunk] tile conv2 reads.  Halo pixels
            // outside the canvas are conv2's SAME padding: zeros, not conv1(0).
            static_assert(!c_fuse1 || NW == 4, "the fused first layer is written for four waves");
            constexpr int HR = TH + 4, HC = TW + 4;          // halo tile of conv2 (k5, stride 1)
            constexpr int UR = HR + 4, UCB = 96;             // uint8 tile rows; bytes per copy row (48 px)
            char* f1 = smem + a.lds_f1_off;                  // [2 copies][UR][48] bf16
            // lane = tile column (48 per copy row, 40 carry data), wave = row phase: rows wave, wave+4, ...
            // The row part of every address is wave-uniform (SGPR base + per-lane column offset), the
            // LDS addresses are one per-lane base plus immediates; all loads are issued before first use.
            constexpr int NU = UR / 4;                       // 6 byte loads per lane
            static_assert(UR % 4 == 0, "uint8 tile rows must be a multiple of the wave count");
            static_assert(NU <= F1_NU, "first-layer prefetch registers");
            auto load_u8 = [&](int oy, int ox) {
                const int xg = ox - 4 + lane;
                const bool colok = lane < HC + 4 && xg >= 0 && xg < a.f1_W;
                const unsigned xo = colok ? (unsigned)xg : 0u;
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const int y = oy - 4 + wave + 4 * u;                    // wave-uniform
                    const bool rowok = y >= 0 && y < a.f1_H;
                    const uint8_t* rowp = a.f1_img + (size_t)(rowok ? y : 0) * (size_t)a.f1_W;
                    const uint8_t v = rowp[xo];
                    ubf[u] = (colok && rowok) ? (float)v : 0.0f;
                }
            };
            if (first_tile) load_u8(oy0, ox0);      // later tiles: requested one tile ahead (below)
            if (lane < 48) {
                char* pA = f1 + wave * UCB + lane * 2;
#pragma unroll
                for (int u = 0; u < NU; ++u) *(uint16_t*)(pA + u * 4 * UCB) = d_f2bf(ubf[u] * 0.00392156886f);
                if (lane >= 1) {
#pragma unroll
                    for (int u = 0; u < NU; ++u) *(uint16_t*)(pA + UR * UCB - 2 + u * 4 * UCB) = d_f2bf(ubf[u] * 0.00392156886f);
                }
            }
            if constexpr (c_persist) {
                // the next tile's bytes are requested now: their latency hides under this tile's conv1 stage,
                // k-loop and epilogue instead of opening the next trip
                const int nt = tile + (int)gridDim.x;
                if (nt < a.ntiles) {
                    const int ntt = xcd_tile(nt);
                    const int nty = ntt / tiles_x, ntx = ntt - nty * tiles_x;
                    load_u8(nty * TH, ntx * TW);
                }
            }
            bf16x8 w1[2][2];
#pragma unroll
            for (int s1 = 0; s1 < 2; ++s1)
#pragma unroll
                for (int q = 0; q < 2; ++q) w1[s1][q] = *(const bf16x8*)(a.f1_wpk + ((size_t)(s1 * 2 + q) * 64 + lane) * 8);
            const float4 b1a = *(const float4*)(a.f1_bias + 4 * g), b1b = *(const float4*)(a.f1_bias + 16 + 4 * g);
            lds_barrier();   // copies visible; does not drain the weight DMAs already in flight
            // 20 halo rows x 36 halo columns = per row two full 16-pixel tiles + 4 columns; the remainder
            // columns of four rows form one more tile (5 in all).  Full tiles: the row is wave-uniform
            // (wave + 4j), so every address is a per-lane constant plus a scalar / immediate; remainder
            // tiles carry their row in the lane.  Each wave runs 10 full + 2 remainder tiles (the second
            // remainder slot is real only for wave 0 -- the others redo theirs, same values).
            static_assert(HR % 4 == 0 && HC == 36, "conv1 tile walk: halo rows in fours, two 16-pixel tiles + 4 columns per row");
            auto c1_tile = [&](const char* src, char* dst, bool inc) {
                // the bias is the accumulators' start value (one rounding order for every first-layer kernel of the engine)
                f32x4 z0 = f32x4{b1a.x, b1a.y, b1a.z, b1a.w}, z1 = f32x4{b1b.x, b1b.y, b1b.z, b1b.w};
#pragma unroll
                for (int s1 = 0; s1 < 2; ++s1) {
                    // kernel row ky = g (first k-step) or 4 (second; rows past the kernel carry zero weights)
                    const uint32_t* rp = (const uint32_t*)(src + (s1 == 0 ? g : 4) * UCB);
                    const uint4 xv = make_uint4(rp[0], rp[1], rp[2], rp[3]);
                    z0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[s1][0], __builtin_bit_cast(bf16x8, xv), z0, 0, 0, 0);
                    z1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[s1][1], __builtin_bit_cast(bf16x8, xv), z1, 0, 0, 0);
                }
                // round, then ReLU on the packed pairs (the MFMA results must be read by instructions the compiler's hazard
                // recogniser sees -- an inline-asm v_max straight on them reads before they have landed)
                const uint32_t fl2 = a.f1_relu ? 0u : 0x80008000u;
                uint2 pa = make_uint2(relu_pk_bf16(pk_bf16(z0[0], z0[1]), fl2), relu_pk_bf16(pk_bf16(z0[2], z0[3]), fl2));
                uint2 pb = make_uint2(relu_pk_bf16(pk_bf16(z1[0], z1[1]), fl2), relu_pk_bf16(pk_bf16(z1[2], z1[3]), fl2));
                if (!inc) { pa = make_uint2(0, 0); pb = make_uint2(0, 0); }   // halo outside the canvas = conv2's zero padding
                *(uint2*)(dst + g * 8) = pa;                  // couts 4g .. 4g+3
                if (g < 2) *(uint2*)(dst + 32 + g * 8) = pb;  // couts 16..19 (g = 0), zero pad 20..23 (g = 1)
            };
            // per-lane constants of the two full column tiles and of the remainder tile
            const char* srcF[2]; char* dstF[2]; bool colF[2];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int hx = ct * 16 + p16, gx = ox0 - 2 + hx;
                srcF[ct] = f1 + (hx & 1) * (UR * UCB) + (hx & ~1) * 2 + wave * UCB;
                dstF[ct] = in_t + hx * c_PS2 + wave * a.row_pitch;
                colF[ct] = gx >= 0 && gx < a.Win;
            }
            const int step_dst = 4 * a.row_pitch;
#pragma unroll
            for (int j = 0; j < HR / 4; ++j) {
                const int gy = oy0 - 2 + wave + 4 * j;
                const bool rowok = gy >= 0 && gy < a.Hin;
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) c1_tile(srcF[ct] + j * 4 * UCB, dstF[ct] + j * step_dst, rowok && colF[ct]);
            }
            {
                const int hxR = 32 + (p16 & 3), hrL = p16 >> 2, gxR = ox0 - 2 + hxR;
                const bool colR = gxR >= 0 && gxR < a.Win;
                constexpr int NREM = HR / 4;                     // remainder tiles (4 rows x 4 columns each)
#pragma unroll
                for (int jj = 0; jj < (NREM + 3) / 4; ++jj) {
                    const int q = min(wave + 4 * jj, NREM - 1);  // surplus slots redo the last tile (same values)
                    const int hr = 4 * q + hrL, gy = oy0 - 2 + hr;
                    c1_tile(f1 + (hxR & 1) * (UR * UCB) + (hxR & ~1) * 2 + hr * UCB, in_t + hr * a.row_pitch + hxR * c_PS2,
                            colR && gy >= 0 && gy < a.Hin);
                }
            }
        } else if (!(c_dbg & 1) && (!c_inrelu || !getenv_vgpr_inrelu) && nbi == 0) {
            // LDS-DMA staging (buffer_load_dwordx4 ... lds): lane L of one instruction fills
            // the 16-byte slot (j*64 + L) of a tile row, slot = pixel*sigma + chunk.  A lane
            // whose pixel lies outside the image, or whose slot is row padding (chunk >= nc),
            // uses an out-of-range offset: the hardware writes zeros there without touching
            // memory (SAME padding for free); lanes past the row end are masked off.  No VGPR
            // round trip, no ds_write, and every row of the wave is in flight at once.
            // Everything that depends only on the column is computed once per block.
            const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.src0, 0, a.bytes0, 0x00020000);
            const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.src1 ? a.src1 : a.src0), 0, a.src1 ? a.bytes1 : 0u, 0x00020000);
            constexpr int JMAX = FIXED ? ((((TW - 1) * (ST_ > 0 ? ST_ : 1) + (KS_ > 0 ? KS_ : 1)) * (SG_ > 0 ? SG_ : 1) + 63) >> 6) : 8;
            constexpr unsigned OOB = 0xfffffff0u;
            const int row_slots = c_TWH * c_sigma;
            const int J = (row_slots + 63) >> 6;
            const unsigned inv = 65536u / (unsigned)c_sigma + 1u;
            const bool any0 = (c0 < a.nch0), any1 = (c0 + nc > a.nch0);
            unsigned col[JMAX];   // column part of the byte offset, or OOB
            int kind[JMAX];       // 0: source 0 (or zero fill), 1: source 1, -1: lane past the row end
#pragma unroll
            for (int j = 0; j < JMAX; ++j) {
                const int sl = j * 64 + lane;
                const int ps = (int)(((unsigned)sl * inv) >> 16), cc = sl - ps * c_sigma;   // pixel slot of the row, chunk
                // stride 2: slots [0, ceil(TWH/2)) hold the even tile columns, the rest the odd ones -- under a tap kx the 16
                // lanes of a fragment read then walk consecutive slots of plane kx & 1 instead of every other pixel (whose
                // 2 * sigma * 16-byte pitch put four to eight lanes on the same banks)
                const int px = c_stride == 2 ? (ps < ((c_TWH + 1) >> 1) ? 2 * ps : 2 * (ps - ((c_TWH + 1) >> 1)) + 1) : ps;
                const int ix = ix0 + px, gc = c0 + cc;
                const bool okc = cc < nc && ix >= 0 && ix < a.Win;
                const bool is1 = okc && gc >= a.nch0;
                col[j] = !okc ? OOB
                              : (is1 ? (unsigned)((ix >> c_up1) * a.nch1 + (gc - a.nch0)) * 16u
                                     : (unsigned)((ix >> c_up0) * a.nch0 + gc) * 16u);
                kind[j] = sl >= row_slots ? -1 : (is1 ? 1 : 0);
            }
            for (int py = wave; py < c_THH; py += NW) {
                const int iy = iy0 + py;
                const bool rowv = (iy >= 0 && iy < a.Hin);
                const unsigned rb0 = rowv ? (unsigned)(iy >> c_up0) * (unsigned)W0 * (unsigned)(a.nch0 * 16) : OOB;
                const unsigned rb1 = rowv ? (unsigned)(iy >> c_up1) * (unsigned)W1 * (unsigned)(a.nch1 * 16) : OOB;
                char* drow = in_t + py * a.row_pitch;
#pragma unroll
                for (int j = 0; j < JMAX; ++j) {
                    if (j < J) {
                        __attribute__((address_space(3))) void* dl = (__attribute__((address_space(3))) void*)(drow + j * 1024);
                        // OOB + anything stays far out of range (32-bit wrap cannot reach a valid offset
                        // because tensors are < 2 GiB): use saturating select instead of an add on OOB
                        if (any0 || !any1) {
                            const unsigned o0 = (col[j] == OOB || !rowv) ? OOB : rb0 + col[j];
                            if (kind[j] == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, dl, 16, o0, 0, 0, 0);
                        }
                        if (any1) {
                            const unsigned o1 = (col[j] == OOB || !rowv) ? OOB : rb1 + col[j];
                            if (kind[j] == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, dl, 16, o1, 0, 0, 0);
                        }
                    }
                }
            }
        } else if (!(c_dbg & 1) && nbi == 0) {
            const int row_items = c_TWH * nc;
            const int J = (row_items + 63) >> 6;          // loads per lane per row
            const int rows_w = (c_THH - wave + NW - 1) / NW;   // rows of this wave
            const int total = rows_w * J;
            const unsigned inv = 65536u / (unsigned)nc + 1u;
            for (int e0 = 0; e0 < total; e0 += STAGE_SLOTS) {
                uint4 v[STAGE_SLOTS];
                int dsto[STAGE_SLOTS];
#pragma unroll
                for (int u = 0; u < STAGE_SLOTS; ++u) {
                    const int e = e0 + u;              // wave-uniform
                    v[u] = make_uint4(0, 0, 0, 0);
                    dsto[u] = -1;
                    if (e < total) {
                        const int r = e / J, j = e - r * J;
                        const int py = wave + NW * r;
                        const int iy = iy0 + py;
                        const int i = j * 64 + lane;
                        if (i < row_items) {
                            const int px = (int)(((unsigned)i * inv) >> 16), cc = i - px * nc;
                            const int ix = ix0 + px;
                            const int ps = c_stride == 2 ? ((px & 1) ? ((c_TWH + 1) >> 1) + (px >> 1) : (px >> 1)) : px;   // de-interleaved slot
                            dsto[u] = py * a.row_pitch + ps * c_PS2 + cc * 16;
                            if (iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win) {
                                const int gc = c0 + cc;
                                const uint16_t* sp = gc < a.nch0
                                    ? a.src0 + ((size_t)(iy >> c_up0) * W0 + (ix >> c_up0)) * (a.nch0 * 8) + gc * 8
                                    : a.src1 + ((size_t)(iy >> c_up1) * W1 + (ix >> c_up1)) * (a.nch1 * 8) + (gc - a.nch0) * 8;
                                v[u] = *(const uint4*)sp;
                            }
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < STAGE_SLOTS; ++u)
                    if (dsto[u] >= 0) *(uint4*)(in_t + dsto[u]) = c_inrelu ? relu_bf16x8(v[u]) : v[u];
            }
        }
        if (first_tile)
#pragma unroll
        for (int u = 0; u < TU; ++u) { const int i = tid + u * NTHR; if (i < ks * 4) tab_l[i] = tabv[u]; }
        if (b == 0) { PSEG_STAMP(3) }

        const int groups_b = (ks + a.GK - 1) / a.GK;
        for (int lg = 0; lg < groups_b; ++lg, ++gq) {
            // group gq has landed once at most the loads of the younger groups are pending
            const int younger = min(D - 1, G - 1 - gq);
            long long tw0 = 0;
            if (c_trace) tw0 = __builtin_amdgcn_s_memtime();
            wait_vmcnt_le(((c_dbg & 8) || lg == 0) ? 0 : L * (younger > 0 ? younger : 0));
            lds_barrier();
            if (c_trace) { const long long t1 = __builtin_amdgcn_s_memtime(); acc_wait += t1 - tw0; tw0 = t1; }
            if (b == 0 && lg == 0) { PSEG_STAMP(4) }
            if (c_inrelu && !c_relu_frag && lg == 0 && !getenv_vgpr_inrelu) {
                // pre-activation ReLU (res_unet) on the tile the DMA just delivered: one in-place pass over
                // the LDS tile (zeros of the padding stay zeros), then every wave may read it
                const int tile_bytes = c_THH * a.row_pitch;
                for (int i = tid * 16; i < tile_bytes; i += NTHR * 16) {
                    uint4* q = (uint4*)(in_t + i);
                    *q = relu_bf16x8(*q);
                }
                lds_barrier();
            }
            if (gq + D < G) stage_w(gq + D, slot_n);  // reuses the slot of group gq-1: free after the barrier
            if (c_trace) acc_issue += __builtin_amdgcn_s_memtime() - tw0;
            slot_n = slot_n + 1 == a.NB ? 0 : slot_n + 1;
            const int n = min(a.GK, ks - lg * a.GK);
            const char* wb = w_t + slot_c * WBUF + lane * 16;
            slot_c = slot_c + 1 == a.NB ? 0 : slot_c + 1;
            const int* tb = tab_l + lg * a.GK * 4 + g;
            // two-stage software pipeline: the fragments of k-step s+1 are in flight while the
            // MFMAs of k-step s issue (A/B are static register sets; index clamped at the tail).
            bf16x8 xa[MT], wa[NT], xb[MT], wbq[NT];
            // (read order = the order the next step's MFMAs need them: cout tile 0, the pixel tiles, the other cout tiles)
#define PSEG_LOAD(XF, WF, S, OFF)                                                                \
            {                                                                                    \
                const int s_ = (S) < n ? (S) : n - 1;                                            \
                WF[0] = *(const bf16x8*)(wb + (s_ * NT) * 1024);                                 \
                _Pragma("unroll") for (int m = 0; m < MT; ++m)                                   \
                    XF[m] = *(const bf16x8*)(in_t + pixbase[m] + OFF);                           \
                _Pragma("unroll") for (int t = 1; t < NT; ++t)                                   \
                    WF[t] = *(const bf16x8*)(wb + (s_ * NT + t) * 1024);                         \
            }
#define PSEG_TAB(S) tb[((S) < n ? (S) : n - 1) * 4]
#define PSEG_MMA(XF, WF)                                                                         \
            _Pragma("unroll") for (int t = 0; t < NT; ++t)                                       \
                _Pragma("unroll") for (int m = 0; m < MT; ++m)                                   \
                    acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WF[t], XF[m], acc[m][t], 0, 0, 0);
            // One scheduling region per k-step: its MT x NT MFMAs with the fragment reads of the NEXT k-step issued between them
            // (one read and one address add behind each MFMA).  Issuing all reads first and then all MFMAs -- the earlier
            // form -- left the matrix pipe idle while the wave issued its reads; co-resident waves filled only part of that.
#define PSEG_INTERLEAVE                                                                          \
            _Pragma("unroll") for (int q_ = 0; q_ < MT * NT; ++q_) {                              \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                \
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                \
                __builtin_amdgcn_sched_group_barrier(0x002, c_relu_frag ? 2 : 1, 0);              \
            }
#define PSEG_RELU(XF)                                                                            \
            if constexpr (c_relu_frag) {                                                         \
                _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                 \
                    const uint4 q_ = __builtin_bit_cast(uint4, XF[m]);                           \
                    XF[m] = __builtin_bit_cast(bf16x8, make_uint4(relu_pk_bf16(q_.x, 0u), relu_pk_bf16(q_.y, 0u), relu_pk_bf16(q_.z, 0u), relu_pk_bf16(q_.w, 0u))); \
                }                                                                                \
            }
            int offa = PSEG_TAB(0), offb = PSEG_TAB(1);
            PSEG_LOAD(xa, wa, 0, offa)
            PSEG_RELU(xa)
            int s = 0;
            for (; s + 2 <= n; s += 2) {
                __builtin_amdgcn_sched_barrier(0);
                offa = PSEG_TAB(s + 2);                     // the table read goes first: the next region opens with its result
                PSEG_LOAD(xb, wbq, s + 1, offb)
                PSEG_MMA(xa, wa)
                PSEG_RELU(xb)
                PSEG_INTERLEAVE
                __builtin_amdgcn_sched_barrier(0);
                offb = PSEG_TAB(s + 3);
                PSEG_LOAD(xa, wa, s + 2, offa)
                PSEG_MMA(xb, wbq)
                PSEG_RELU(xa)
                PSEG_INTERLEAVE
            }
            __builtin_amdgcn_sched_barrier(0);
            if (n & 1) { PSEG_MMA(xa, wa) }
#undef PSEG_RELU
#undef PSEG_INTERLEAVE
#undef PSEG_TAB
#undef PSEG_LOAD
#undef PSEG_MMA
        }
    }

    PSEG_STAMP(5)
    if (c_trace && tid == 0) {
        c_trace[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 12 + 8] = (unsigned long long)acc_wait;
        c_trace[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 12 + 9] = (unsigned long long)acc_issue;
    }
    do {
    // ---- epilogue -----------------------------------------------------------------------------
    // D layout: lane holds pixel (lane & 15) x couts 4*(lane>>4) .. +3 of each 16x16 tile.
    // The inline-asm max / DPP instructions below may be the first readers of accumulator registers, and the compiler's
    // hazard recogniser does not look inside inline asm: let the last MFMAs of the k-loop (8 passes) retire first.
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
    if (c_dbg & 4) { if (acc[0][0][0] == 123.456f) a.dst[0] = 1; return; }
    if constexpr (NT == 4 || NT == 8) {
        if (c_tail) {
            // Fused tail.  Stage 1 above produced, per half-resolution pixel, the four output
            // pixels' deconv channels: tiles (2ab, 2ab+1) hold co 0..31 of sub-pixel ab (CoP = 32).
            // A 16x16 accumulator tile has the pixel on the lane and four channels in registers,
            // i.e. it already is a B operand (k x pixel) of the next MFMA: two tiles give the 8
            // k-values a lane needs (k = 8g+j <-> co = 4g+j for j<4, 16+4g+(j-4) otherwise; the
            // host packs the logits weights in that k order).  The skip channels come straight
            // from global memory as a second B fragment.  D2[class][pixel] -> argmax / softmax.
            const bf16x8 wa = *(const bf16x8*)(a.tail_wa + lane * 8);
            const bf16x8 wbk = *(const bf16x8*)(a.tail_wb + lane * 8);
            const float4 lb = *(const float4*)(a.tail_bias + 4 * g);
            const int C = a.tail_C;
            constexpr int NAB = NT / 2;      // sub-pixels handled by this N block
            const int ab0 = nb * NAB;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int hy = oy0 + wave * (MT / 2) + (m >> 1), hx = ox0 + (m & 1) * 16 + p16;
#pragma unroll
                for (int abl = 0; abl < NAB; ++abl) {
                    const int ab = ab0 + abl;
                    const int y = 2 * hy + (ab >> 1), x = 2 * hx + (ab & 1);
                    const bool inb = (y < a.H0 && x < a.W0);
                    const f32x4 t0 = acc[m][2 * abl], t1 = acc[m][2 * abl + 1];
                    uint4 dq;
                    dq.x = pk_bf16(t0[0], t0[1]);        // (the deconv bias was the accumulators' start value)
                    dq.y = pk_bf16(t0[2], t0[3]);
                    dq.z = pk_bf16(t1[0], t1[1]);
                    dq.w = pk_bf16(t1[2], t1[3]);
                    f32x4 z = f32x4{lb.x, lb.y, lb.z, lb.w};
                    z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, __builtin_bit_cast(bf16x8, dq), z, 0, 0, 0);
                    if (a.skip)
                        z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wbk, __builtin_bit_cast(bf16x8, skf[m][abl]), z, 0, 0, 0);
                    // argmax over classes 4g+r (first maximum wins), combined across the four g
                    float bv = -3.4e38f;
                    int bi = 0x7fffffff;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int c = 4 * g + r;
                        const bool take = (c < C) & (z[r] > bv);
                        bv = take ? z[r] : bv;
                        bi = take ? c : bi;
                    }
                    if (C > 4) {   // classes beyond the g = 0 lanes: combine across the four lane groups
#pragma unroll
                        for (int sh = 16; sh <= 32; sh <<= 1) {
                            const float ov = __shfl_xor(bv, sh);
                            const int oi = __shfl_xor(bi, sh);
                            const bool take = (ov > bv) | ((ov == bv) & (oi < bi));
                            bv = take ? ov : bv;
                            bi = take ? oi : bi;
                        }
                    }
                    const size_t p = (size_t)y * a.W0 + x;
                    if (inb) {
                        if (g == 0) {
                            if (a.out_labels_u8) a.out_labels_u8[p] = (uint8_t)bi;
                            if (a.out_labels) a.out_labels[p] = bi;
                        }
                        if (a.out_logits)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (4 * g + r < C) a.out_logits[p * C + 4 * g + r] = z[r];
                    }
                    if (a.out_probs) {
                        float ex[4], sum = 0.f;
#pragma unroll
                        for (int r = 0; r < 4; ++r) { ex[r] = (4 * g + r < C) ? expf(z[r] - bv) : 0.f; sum += ex[r]; }
                        if (C > 4) {
                            sum += __shfl_xor(sum, 16);
                            sum += __shfl_xor(sum, 32);
                        }
                        if (inb)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (4 * g + r < C) a.out_probs[p * C + 4 * g + r] = ex[r] / sum;
                    }
                }
            }
            PSEG_STAMP(6)
            break;
        }
    }
    if (c_deconv) {
        // Conv2DTranspose k2 s2: n = ab*CoP + co; scatter to (2y + a, 2x + b).  8-byte stores.
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int y = oy0 + wave * (MT / 2) + (m >> 1), x = ox0 + (m & 1) * 16 + p16;
            if (y >= a.Hout || x >= a.Wout) continue;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int n = (nb * NT + t) * 16 + 4 * g;
                const int ab = n / a.CoP, co = n - ab * a.CoP;
                if (ab >= 4) continue;
                float v0 = acc[m][t][0], v1 = acc[m][t][1];
                float v2 = acc[m][t][2], v3 = acc[m][t][3];
                if (a.relu) {
                    v0 = v0 > 0.f ? v0 : 0.f; v1 = v1 > 0.f ? v1 : 0.f;
                    v2 = v2 > 0.f ? v2 : 0.f; v3 = v3 > 0.f ? v3 : 0.f;
                }
                const uint2 pk = make_uint2(pk_bf16(v0, v1),
                                            pk_bf16(v2, v3));
                const size_t o = ((size_t)(2 * y + (ab >> 1)) * (2 * a.Wout) + (2 * x + (ab & 1))) * (a.nch_out * 8) + co;
                *(uint2*)(a.dst + o) = pk;
            }
        }
        break;
    }

    if constexpr (FIXED && (FL_ & FL_LOGITS) != 0 && NT == 4) {
        // Fused logits (unet tail): the 64 couts of a pixel sit in four accumulator tiles, i.e. they already
        // are the B operands (k x pixel) of two logits MFMAs (k = 8g+j <-> cout 16(2q) + 4g + j for j < 4,
        // 16(2q+1) + 4g + (j-4) otherwise; the host packs the logits kernel in that order).  +bias, ReLU and
        // the bf16 rounding of the layer output happen in registers; the 64-channel tensor is never stored.
        const bf16x8 wq0 = *(const bf16x8*)(a.tail_wa + lane * 8), wq1 = *(const bf16x8*)(a.tail_wb + lane * 8);
        const float4 lb = *(const float4*)(a.tail_bias + 4 * g);
        const int C = a.tail_C;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int y = oy0 + wave * (MT / 2) + (m >> 1), x = ox0 + (m & 1) * 16 + p16;
            const bool inb = y < a.H0 && x < a.W0;
            uint32_t pk[NT][2];
            uint2 adr[NT];
            if (c_add) {   // residual operand of the block (res_unet): same pixel, this lane's four couts per tile
                const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)a.add, 0, a.dst_bytes, 0x00020000);
                const bool inp = y < a.Hout && x < a.Wout;
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    adr[t] = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(
                        ra, inp ? (unsigned)(y * a.Wout + x) * (unsigned)(a.nch_out * 16) + (unsigned)(t * 16 + 4 * g) * 2u : 0xfffffff0u, 0, 0));
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                float v0 = acc[m][t][0], v1 = acc[m][t][1];
                float v2 = acc[m][t][2], v3 = acc[m][t][3];
                if (c_add) {
                    v0 += d_bf2f((uint16_t)(adr[t].x & 0xffff)); v1 += d_bf2f((uint16_t)(adr[t].x >> 16));
                    v2 += d_bf2f((uint16_t)(adr[t].y & 0xffff)); v3 += d_bf2f((uint16_t)(adr[t].y >> 16));
                }
                if (a.relu) { v0 = vmax(v0, 0.f); v1 = vmax(v1, 0.f); v2 = vmax(v2, 0.f); v3 = vmax(v3, 0.f); }
                pk[t][0] = pk_bf16(v0, v1);
                pk[t][1] = pk_bf16(v2, v3);
            }
            f32x4 z = f32x4{lb.x, lb.y, lb.z, lb.w};
            z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq0, __builtin_bit_cast(bf16x8, make_uint4(pk[0][0], pk[0][1], pk[1][0], pk[1][1])), z, 0, 0, 0);
            z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq1, __builtin_bit_cast(bf16x8, make_uint4(pk[2][0], pk[2][1], pk[3][0], pk[3][1])), z, 0, 0, 0);
            float bv = -3.4e38f;
            int bi = 0x7fffffff;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 4 * g + r;
                const bool take = (c < C) & (z[r] > bv);
                bv = take ? z[r] : bv;
                bi = take ? c : bi;
            }
            if (C > 4) {
#pragma unroll
                for (int sh = 16; sh <= 32; sh <<= 1) {
                    const float ov = __shfl_xor(bv, sh);
                    const int oi = __shfl_xor(bi, sh);
                    const bool take = (ov > bv) | ((ov == bv) & (oi < bi));
                    bv = take ? ov : bv;
                    bi = take ? oi : bi;
                }
            }
            const size_t p = (size_t)y * a.W0 + x;
            if (inb) {
                if (g == 0) {
                    if (a.out_labels_u8) a.out_labels_u8[p] = (uint8_t)bi;
                    if (a.out_labels) a.out_labels[p] = bi;
                }
                if (a.out_logits)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (4 * g + r < C) a.out_logits[p * C + 4 * g + r] = z[r];
            }
            if (a.out_probs) {
                float ex[4], sum = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) { ex[r] = (4 * g + r < C) ? expf(z[r] - bv) : 0.f; sum += ex[r]; }
                if (C > 4) {
                    sum += __shfl_xor(sum, 16);
                    sum += __shfl_xor(sum, 32);
                }
                if (inb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (4 * g + r < C) a.out_probs[p * C + 4 * g + r] = ex[r] / sum;
            }
        }
        break;
    }

    if constexpr (FIXED && (FL_ & FL_DQ) != 0) {
        // Conv2DTranspose k2 s2 (ReLU) behind this conv, on its accumulators: a 16x16 accumulator tile has the pixel on the
        // lane and four channels in registers, so two tiles are one B operand (k x pixel) of the next MFMA -- the conv's 80
        // output channels (bias was the start value; ReLU, bf16 rounding here) are three k-steps, the third half empty.  Per
        // sub-pixel ab: D2[cout][pixel] = bias2 + W2[ab] . d1(pixel), ReLU, bf16, stored at (2y + ab / 2, 2x + ab % 2).
        // The 1/8-resolution tensor never exists in memory and the transposed conv's own launch is gone.
        static_assert(!(FL_ & FL_DQ) || (NT == 5 && MT == 4), "the fused transposed conv is written for 80 channels on 8 x 32 tiles");
        uint4 bq[MT][3];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            uint32_t pk[NT][2];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                pk[t][0] = pk_bf16(acc[m][t][0], acc[m][t][1]);
                pk[t][1] = pk_bf16(acc[m][t][2], acc[m][t][3]);
                if (a.relu) { pk[t][0] = relu_pk_bf16(pk[t][0], 0u); pk[t][1] = relu_pk_bf16(pk[t][1], 0u); }
            }
            bq[m][0] = make_uint4(pk[0][0], pk[0][1], pk[1][0], pk[1][1]);
            bq[m][1] = make_uint4(pk[2][0], pk[2][1], pk[3][0], pk[3][1]);
            bq[m][2] = make_uint4(pk[4 % NT][0], pk[4 % NT][1], 0u, 0u);
        }
        const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)a.dq_dst, 0, a.dq_bytes, 0x00020000);
        const int Cs2 = a.dq_nch * 8, W2 = 2 * a.Wout;
        constexpr unsigned OOBQ = 0xfffffff0u;
        unsigned pxo[MT];     // byte offset of the (2y, 2x) output pixel of this lane's conv pixel, or OOB
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int y = oy0 + wave * (MT / 2) + (m >> 1), x = ox0 + (m & 1) * 16 + p16;
            pxo[m] = (y < a.Hout && x < a.Wout) ? (unsigned)((2 * y) * W2 + 2 * x) * (unsigned)(Cs2 * 2) : OOBQ;
        }
        // eight groups (sub-pixel ab, pair of cout tiles th): the six A fragments of group q + 1 are requested before group q is
        // multiplied (they come from L2: 48 KB shared by every workgroup)
        bf16x8 wf[2][2][3];
        auto fetch_w = [&](int q, bf16x8 (&w)[2][3]) {
            const int ab = q >> 1, th = q & 1;
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int s3 = 0; s3 < 3; ++s3)
                    w[tt][s3] = *(const bf16x8*)(a.dq_w + ((size_t)((ab * 4 + 2 * th + tt) * 3 + s3) * 64 + lane) * 8);
        };
        float4 b2[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) b2[t] = *(const float4*)(a.dq_bias + t * 16 + 4 * g);
        fetch_w(0, wf[0]);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int ab = q >> 1, th = q & 1;
            if (q + 1 < 8) fetch_w(q + 1, wf[(q + 1) & 1]);
            const unsigned abo = (unsigned)((ab >> 1) * W2 + (ab & 1)) * (unsigned)(Cs2 * 2);
            f32x4 z[MT][2];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int m = 0; m < MT; ++m) z[m][tt] = f32x4{b2[2 * th + tt].x, b2[2 * th + tt].y, b2[2 * th + tt].z, b2[2 * th + tt].w};
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        z[m][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[q & 1][tt][s3], __builtin_bit_cast(bf16x8, bq[m][s3]), z[m][tt], 0, 0, 0);
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int n = (2 * th + tt) * 16 + 4 * g;
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    uint2 pk = make_uint2(pk_bf16(z[m][tt][0], z[m][tt][1]), pk_bf16(z[m][tt][2], z[m][tt][3]));
                    if (a.dq_relu) pk = make_uint2(relu_pk_bf16(pk.x, 0u), relu_pk_bf16(pk.y, 0u));
                    const unsigned o = (pxo[m] == OOBQ || n >= Cs2) ? OOBQ : pxo[m] + abo + (unsigned)n * 2u;
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, pk), rq, o, 0, 0);
                }
            }
        }
        break;
    }
    // Direct stores: lane (p16, g) owns couts 4g..4g+3 of pixel p16 in every 16x16 tile, i.e.
    // 8 contiguous bytes of the NHWC row; the cout tiles / four g of a pixel complete its line
    // across consecutive store instructions.  Bounds are enforced by the buffer descriptors (an
    // out-of-range offset drops the store), so the epilogue is branch-free.  The fused 2x2
    // max-pool is taken on the float32 values in registers (rows m / m+2 are vertical
    // neighbours in the same lane, x-neighbours are lanes p16 ^ 1); rounding is monotone, so it
    // equals pooling the stored bf16 tensor.
    const int CsO = a.nch_out * 8;
    constexpr unsigned OOBS = 0xfffffff0u;
    constexpr bool c_skiplog = FIXED && (FL_ & FL_SKIPLOG) != 0 && NT == 2;
    uint2 pks[2][c_skiplog ? MT : 1];
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)a.dst, 0, a.dst_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)(c_pool ? a.pool_dst : a.dst), 0, c_pool ? a.pool_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd2 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.dst2 ? a.dst2 : a.dst), 0, a.dst2 ? a.dst_bytes : 0u, 0x00020000);
    unsigned pixoff[MT];   // byte offset of this lane's pixel in dst, or OOBS
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int y = oy0 + wave * (MT / 2) + (m >> 1), x = ox0 + (m & 1) * 16 + p16;
        pixoff[m] = (y < a.Hout && x < a.Wout) ? (unsigned)(y * a.Wout + x) * (unsigned)(CsO * 2) : OOBS;
    }
    // Stores through LDS (plain conv instances).  A direct store instruction carries 16 pixels x 32 bytes: sixteen cache lines
    // for half a KiB, and switching the epilogue off showed what that costs -- unet's full-resolution layers 339 -> 201 us, the
    // quarter-resolution layers of fcn_skip 5-12 us each (every workgroup of a one-round kernel stores at the same moment).  The
    // waves therefore write their packed tiles into an LDS patch laid out like the tensor ([row][pixel][this N block's NT * 32
    // bytes], pixel pitch + 8 bytes: conflict-free 8-byte writes) -- the input tile and the weight ring are free once every wave
    // has left the k-loop -- and read it back as 16-byte pieces of consecutive addresses: a store instruction is then 1 KiB of
    // whole 128-byte lines.  Same bytes, same bounds (descriptor range check).
    constexpr bool c_lst = FIXED && MODE_ == MODE_CONV && MT == 4 && (FL_ & (FL_ADD | FL_SKIPLOG | FL_LOGITS | FL_DQ | FL_PERSIST | FL_FUSE1)) == 0;
    constexpr int LST_PP = NT * 32 + 8;                                   // patch bytes per pixel
    const bool lst = c_lst && !a.dst2 && a.dst_bytes != 0u && !(a.dbg & 32);
    char* const patch = smem + wave * ((MT / 2) * TW * LST_PP);           // this wave's rows
    if (lst) lds_barrier();
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = (nb * NT + t) * 16 + 4 * g;
        const unsigned noff = n < CsO ? (unsigned)n * 2u : OOBS;
        float v[MT][4];
        // residual operand (res_unet): the MT loads of this cout tile are all requested before the first use
        uint2 adr[MT];
        if (c_add) {
            const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)a.add, 0, a.dst_bytes, 0x00020000);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const unsigned o = (pixoff[m] == OOBS || noff == OOBS) ? OOBS : pixoff[m] + noff;
                adr[m] = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(ra, o, 0, 0));
            }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            v[m][0] = acc[m][t][0]; v[m][1] = acc[m][t][1]; v[m][2] = acc[m][t][2]; v[m][3] = acc[m][t][3];
            const unsigned o = (pixoff[m] == OOBS || noff == OOBS) ? OOBS : pixoff[m] + noff;
            if (c_add) {
                const uint2 ad = adr[m];
                v[m][0] += d_bf2f((uint16_t)(ad.x & 0xffff)); v[m][1] += d_bf2f((uint16_t)(ad.x >> 16));
                v[m][2] += d_bf2f((uint16_t)(ad.y & 0xffff)); v[m][3] += d_bf2f((uint16_t)(ad.y >> 16));
            }
            if (a.relu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[m][r] = vmax(v[m][r], 0.0f);
            }
            const uint2 pk = make_uint2(pk_bf16(v[m][0], v[m][1]),
                                        pk_bf16(v[m][2], v[m][3]));
            if constexpr (c_skiplog) pks[t < 2 ? t : 0][m] = pk;     // consumed by the logits MFMA below instead of memory
            else if (lst) {
                *(uint2*)(patch + ((m >> 1) * TW + (m & 1) * 16 + p16) * LST_PP + t * 32 + g * 8) = pk;
            } else {
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, pk), rd, o, 0, 0);
                if (a.dst2) {   // (wave-uniform) the ReLU'd copy the pre-activation readers stage as it is
                    const uint2 pr = make_uint2(relu_pk_bf16(pk.x, 0u), relu_pk_bf16(pk.y, 0u));
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, pr), rd2, o, 0, 0);
                }
            }
        }
        if (c_pool) {
            const int Wo2 = a.Wout >> 1, Ho2 = a.Hout >> 1;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                if ((m >> 1) & 1) continue;  // pairs (m, m+2): rows 2r and 2r+1 of this wave
                float q[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    q[r] = vmax_xor1(vmax(v[m][r], v[m + 2][r]));
                }
                const int y = (oy0 >> 1) + ((wave * (MT / 2) + (m >> 1)) >> 1);
                const int x = (ox0 >> 1) + (((m & 1) * 16 + p16) >> 1);
                const bool ok = !(p16 & 1) && y < Ho2 && x < Wo2 && noff != OOBS;
                const unsigned o = ok ? (unsigned)(y * Wo2 + x) * (unsigned)(CsO * 2) + noff : OOBS;
                const uint2 pk = make_uint2(pk_bf16(q[0], q[1]),
                                            pk_bf16(q[2], q[3]));
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, pk), rp, o, 0, 0);
            }
        }
    }
    if (lst) {
        // read back: piece i = lane + 64 u of this wave's rows = 16 bytes c of pixel px of row r (pieces in address order)
        constexpr int PPX = NT * 2;                                       // 16-byte pieces per pixel
        constexpr int NP = (MT / 2) * TW * PPX;                           // pieces of this wave
        const int nbyte0 = nb * NT * 32;                                  // this N block's first byte inside a pixel
#pragma unroll
        for (int u = 0; u < NP / 64; ++u) {
            const int i = lane + 64 * u;
            const int px = i / PPX, c = i - px * PPX, r = px / TW, xx = px - r * TW;
            const char* sp = patch + px * LST_PP + c * 16;
            const uint2 lo = *(const uint2*)sp, hi = *(const uint2*)(sp + 8);
            const int y = oy0 + wave * (MT / 2) + r, x = ox0 + xx;
            const bool ok = y < a.Hout && x < a.Wout && nbyte0 + c * 16 < CsO * 2;
            const unsigned o = ok ? (unsigned)(y * a.Wout + x) * (unsigned)(CsO * 2) + (unsigned)(nbyte0 + c * 16) : OOBS;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, make_uint4(lo.x, lo.y, hi.x, hi.y)), rd, o, 0, 0);
        }
    }
    if constexpr (c_skiplog) {
        // The skip connection into the logits layer (fcn_skip: conv2 -> logits): this layer's bf16-rounded output is
        // only ever multiplied by the logits kernel, so that product is taken here -- the two accumulator tiles of a
        // pixel are the B operand (k = 8g+j <-> cout 4g+j for j < 4, 16+4g+(j-4) otherwise) -- and 4 (8) floats per
        // pixel leave the kernel instead of the 64-byte tensor row.
        const bf16x8 wq = *(const bf16x8*)(a.tail_wa + lane * 8);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int y = oy0 + wave * (MT / 2) + (m >> 1), x = ox0 + (m & 1) * 16 + p16;
            f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
            z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq, __builtin_bit_cast(bf16x8, make_uint4(pks[0][m].x, pks[0][m].y, pks[1][m].x, pks[1][m].y)), z, 0, 0, 0);
            if (y < a.Hout && x < a.Wout && 4 * g < a.skip_CP)
                *(float4*)(a.skip_logits + ((size_t)y * a.Wout + x) * a.skip_CP + 4 * g) = make_float4(z[0], z[1], z[2], z[3]);
        }
    }
    } while (0);
    if (!c_persist) break;
    tile += gridDim.x;
    if (tile >= a.ntiles) break;
    lds_barrier();   // every wave is out of the k-loop: the input tile (and the first-layer staging area) may be overwritten
    }   // tile loop
    // the next N block reuses the ring and (tail) the staged tile: every wave must be out of the k-loop
    if (nbi + 1 < NBL) lds_barrier();
    }   // N-block loop
    PSEG_STAMP(6)
    if (c_trace && tid == 0) { unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(x)); c_trace[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 12 + 7] = x; }
#undef PSEG_STAMP
}


// ---------------------------------------------------------------------------------------------
// conv1 + conv2 (+ pool, + skip logits) of fcn / fcn_skip as a persistent, WAVE-SPECIALISED kernel: the dominant launch
// of the page (27 % of the FLOPs).  One 512-thread workgroup per CU walks its tiles (16 x 32 output pixels each):
//   * waves 4-7 (one per SIMD) are PRODUCERS: they recompute conv1 (1 -> 20, k5, ReLU) on the next tile's 20 x 36 halo
//     straight from the uint8 page (x/255 fused) into one of TWO conv2 input tiles in LDS.  A producer wave owns five halo
//     rows: it loads the nine page rows they need, keeps its two shifted bf16 copies of them in an LDS area of its own and
//     reads only that area, so the producers never wait for each other;
//   * waves 0-3 (one per SIMD) are CONSUMERS: the implicit-GEMM k-loop of conv2 on the current tile (19 k-steps x 16
//     MFMAs, weights resident in LDS for the whole kernel, two-stage software pipeline) and the epilogue from registers
//     (bias, fused 2x2 max-pool, the logits layer's skip rows by one more MFMA);
//   * ONE s_barrier per tile hands the filled tile to the consumers and the drained one back to the producers.
// In conv_mfma_kernel<.., FL_FUSE1 | FL_PERSIST> every wave did staging, k-loop and epilogue in turn and only the k-loop
// (a third of a tile's life) fed the matrix pipe; here the SIMD's matrix pipe belongs to a wave that does little else
// while the VALU / LDS-write work of the first layer runs beside it on the same SIMD.
// Arithmetic and summation order per output are those of the fused kernel (same k-chunk order, same packing): the two
// produce identical bits (tests/test_bf16_gpu.py).
// ---------------------------------------------------------------------------------------------
#ifndef PSEG_WS_EPI_VALU
#define PSEG_WS_EPI_VALU 2      // epilogue VALU instructions issued behind each MFMA of the consumers' k-loop
#endif
constexpr int WS_ROWP = 1744;     // row pitch of the dense sigma = 3 tile ((36 * 3 + 1) slots, pair_chunks' choice)
constexpr int WS_KSTEPS = 19;     // 25 taps x 3 chunks = 75 k-chunks + 1 dummy
constexpr int WS_KSTEPS_PAIR = 17; // 25 taps x 2 chunks + 15 paired half chunks = 65 k-chunks + 3 dummies
constexpr int WS_TILE0 = 16;      // the tiles start 16 bytes into LDS: a pixel's paired half chunk is stored 8 bytes BEFORE its slot

// PAIRC2: channels 16-19 of a pixel are written twice -- into bytes 32-39 of its own 48-byte slot and into bytes 40-47 of its
// LEFT neighbour's slot -- so that the k-chunk "third chunk under tap kx" also carries tap kx + 1 (see pair_chunks): 17 k-steps.
// FORM: the consumers' tile loop -- 0: one k-loop per tile, epilogue after it (default); 1: two phases of four pixel tiles, the
// epilogue of the previous half between the MFMAs (PSEG_WS_FORM=1); 2: one k-loop per tile with the epilogue of the PREVIOUS
// tile between its MFMAs, two accumulator sets (PSEG_WS_FORM=2).  All three produce the same bits.
template <bool SKIPLOG, bool PAIRC2, int FORM = 0>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv12_ws_kernel(MConv a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    char* const smem = smem_raw + WS_TILE0;
    constexpr int MT = 8, NT = 2, TH = 16, HR = TH + 4, HC = TW + 4, PS2 = 48, ROWP = WS_ROWP, KSTEPS = PAIRC2 ? WS_KSTEPS_PAIR : WS_KSTEPS;
    constexpr int UCB = 96, PR = 9;                       // producer staging: bytes per copy row (48 px), page rows per wave
    constexpr int PRIV = 2 * PR * UCB + 16;               // two shifted copies + a 16-byte sink for stores that carry nothing
    static_assert(HR % 4 == 0 && HR / 4 == 5, "a producer wave owns five halo rows");
    const int TB = a.lds_w_off;                           // bytes of one input tile (HR rows), 16-aligned (host)
    char* const w_t = smem + 2 * TB;
    int* const tab_l = (int*)(w_t + KSTEPS * NT * 1024);
    char* const priv_all = (char*)tab_l + ((KSTEPS * 16 + 15) & ~15);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p16 = lane & 15, g = lane >> 4;
    const int tiles_x = (a.Wout + TW - 1) / TW;
    auto xcd_tile = [&](int t) {
        if (a.xq < 0) return t;
        const int x = t & 7, j = t >> 3;
        return x * a.xq + min(x, a.xr) + j;
    };
    // ---- resident weights + k-chunk table: once per workgroup ------------------------------------------------------
    {
        for (int pc = wave; pc < KSTEPS * NT; pc += 8)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.wpk + (size_t)pc * 512 + lane * 8),
                                             (__attribute__((address_space(3))) void*)(w_t + pc * 1024), 16, 0, 0);
        if (tid < KSTEPS * 4) tab_l[tid] = a.tab_full[tid];
        // bytes no producer ever writes (row padding; with PAIRC2 the right half chunk of the last column) meet zero weights,
        // but must not hold NaN patterns
        for (int i = tid * 16 - WS_TILE0; i < 2 * TB; i += 512 * 16) *(uint4*)(smem + i) = make_uint4(0, 0, 0, 0);
    }
    const int n_my = ((int)a.ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // tiles of this workgroup

    if (wave >= 4) {
        // =============================== PRODUCERS ===============================
        const int pw = wave - 4;
        char* const priv = priv_all + pw * PRIV;
        if ((a.dbg & 0x300) == 0x100) __builtin_amdgcn_s_setprio(1);   // PSEG_WS_PRIO=1: the producers win issue arbitration
        bf16x8 w1[2][2];
#pragma unroll
        for (int s1 = 0; s1 < 2; ++s1)
#pragma unroll
            for (int q = 0; q < 2; ++q) w1[s1][q] = *(const bf16x8*)(a.f1_wpk + ((size_t)(s1 * 2 + q) * 64 + lane) * 8);
        const float4 b1a = *(const float4*)(a.f1_bias + 4 * g), b1b = *(const float4*)(a.f1_bias + 16 + 4 * g);
        const f32x4 zb0 = f32x4{b1a.x, b1a.y, b1a.z, b1a.w}, zb1 = f32x4{b1b.x, b1b.y, b1b.z, b1b.w};
        const uint32_t floor2 = a.f1_relu ? 0u : 0x80008000u;
        typedef float f32x16_t __attribute__((ext_vector_type(16)));
        bf16x8 w3[3];
        f32x16_t zb32;
        {
            const uint16_t* wp = a.f1_wpk32 ? a.f1_wpk32 : a.f1_wpk;       // (the 16x16x32 path never reads w3 / zb32)
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3) w3[s3] = *(const bf16x8*)(wp + ((size_t)s3 * 64 + lane) * 8);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float4 bq = *(const float4*)(a.f1_bias + 8 * v + 4 * (lane >> 5));
                zb32[4 * v] = bq.x; zb32[4 * v + 1] = bq.y; zb32[4 * v + 2] = bq.z; zb32[4 * v + 3] = bq.w;
            }
        }
        // 32x32x16 path, per-lane constants of its two tile types (0: full tile, halo columns 0-31 of row j = immediate; 1: the four
        // remainder columns, row = lane / 4 of the low half, surplus lanes redo row 4): copy offsets of the three k-steps
        // (kernel row 2 s + half; row 5 does not exist: zero weights, any finite data), the slot of the pixel, and where the
        // third / fourth store of the lane goes (half 1 under PAIRC2: the row's padding bytes)
        int c32_src[2][3], c32_d12[2], c32_d3[2], c32_d4[2], c32_hx[2], c32_j[2];
        {
            const int n32 = lane & 31, hh = lane >> 5;
#pragma unroll
            for (int ty = 0; ty < 2; ++ty) {
                const int hx = ty ? 32 + (n32 & 3) : n32, j = ty ? min(n32 >> 2, 4) : 0;
                const int c = (hx & 1) * (PR * UCB) + (hx & ~1) * 2 + j * UCB;
#pragma unroll
                for (int s3 = 0; s3 < 3; ++s3) c32_src[ty][s3] = c + min(2 * s3 + hh, 4) * UCB;
                const int d = hx * PS2 + (5 * pw + j) * ROWP, pad = (5 * pw + j) * ROWP + HC * PS2 + 8;
                c32_d12[ty] = d + 8 * hh;
                c32_d3[ty] = PAIRC2 ? (hh == 0 ? d + 32 : pad) : d + 32 + 8 * hh;
                c32_d4[ty] = hh == 0 ? d - 8 : pad;
                c32_hx[ty] = hx; c32_j[ty] = j;
            }
        }
        const __amdgpu_buffer_rsrc_t rimg = __builtin_amdgcn_make_buffer_rsrc((void*)a.f1_img, 0, (unsigned)((size_t)a.f1_H * a.f1_W), 0x00020000);
        // the wave's 12 first-layer tiles: five rows x two full 16-pixel column tiles (ct = 0, 1), the four remainder
        // columns of rows 0-3 in one tile (type 2) and of row 4 in another (type 3: its surplus lanes redo row 4).
        // Per-lane constants of the four tile types: page-copy offsets of k-step 0 (kernel row g) and k-step 1 (kernel
        // row 4), the tile-buffer offsets of the two 8-byte stores (couts 4g..4g+3; then couts 16..19 / the zero pad
        // 20..23 for g < 2 -- the other lanes repeat their first store, which keeps the code branch-free).
        int srcA[4], srcB[4], dst1[4], dst2[4], hxT[4], jT[4];
        const int sink = (int)(priv - smem) + 2 * PR * UCB + (g & 1) * 8;      // stores of lanes that carry nothing go here
#pragma unroll
        for (int ty = 0; ty < 4; ++ty) {
            const int hx = ty < 2 ? ty * 16 + p16 : 32 + (p16 & 3);
            const int j = ty < 2 ? 0 : (ty == 2 ? (p16 >> 2) : 4);          // row inside the band (full tiles add it as an immediate)
            const int c = (hx & 1) * (PR * UCB) + (hx & ~1) * 2 + j * UCB;
            srcA[ty] = c + g * UCB;
            srcB[ty] = c + 4 * UCB;
            const int d = hx * PS2 + (5 * pw + j) * ROWP;
            dst1[ty] = d + g * 8;
            dst2[ty] = d + 32 + g * 8;
            hxT[ty] = hx; jT[ty] = j;
        }
        const bool low_g = g < 2;
        float ubf[PR];
        auto tile_origin = [&](int i, int& oy, int& ox) {
            const int t = xcd_tile((int)blockIdx.x + i * (int)gridDim.x);
            const int ty = t / tiles_x, tx = t - ty * tiles_x;
            oy = ty * TH; ox = tx * TW;
        };
        auto load_u8 = [&](int oy, int ox) {
            const int xg = ox - 4 + lane, y0 = oy - 4 + 5 * pw;
            if (oy >= 4 && ox >= 4 && oy + TH + 4 <= a.f1_H && ox + TW + 4 <= a.f1_W) {
                // the whole 24 x 40 page tile lies inside the page (every tile but the border ones): one lane offset, the
                // row step rides in the scalar offset
                const unsigned off = (unsigned)(y0 * a.f1_W + ox - 4 + min(lane, HC + 3));
#pragma unroll
                for (int u = 0; u < PR; ++u) ubf[u] = (float)(unsigned char)__builtin_amdgcn_raw_buffer_load_b8(rimg, off, u * a.f1_W, 0);
                return;
            }
            const bool colok = lane < HC + 4 && xg >= 0 && xg < a.f1_W;
#pragma unroll
            for (int u = 0; u < PR; ++u) {
                const int y = y0 + u;                                   // wave-uniform
                const bool ok = colok && y >= 0 && y < a.f1_H;
                const unsigned off = ok ? (unsigned)(y * a.f1_W + xg) : 0xfffffff0u;   // out of range reads 0
                ubf[u] = (float)(unsigned char)__builtin_amdgcn_raw_buffer_load_b8(rimg, off, 0, 0);
            }
        };
        int noy = 0, nox = 0;                               // origin of the tile whose page bytes sit in ubf
        auto fill = [&](int i, int bufoff) {                // tile number i of this workgroup -> smem + bufoff; ubf holds its page bytes
            const int oy0 = noy, ox0 = nox;
            if (lane < 48) {
                char* pA = priv + lane * 2;
#pragma unroll
                for (int u = 0; u < PR; ++u) *(uint16_t*)(pA + u * UCB) = d_f2bf(ubf[u] * 0.00392156886f);
                if (lane >= 1) {
#pragma unroll
                    for (int u = 0; u < PR; ++u) *(uint16_t*)(pA + PR * UCB - 2 + u * UCB) = d_f2bf(ubf[u] * 0.00392156886f);
                }
            }
            if (i + 1 < n_my) {                            // the next tile's bytes: their latency hides under this tile's conv1
                tile_origin(i + 1, noy, nox);
                load_u8(noy, nox);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's own copies (nobody else reads them)
            // a halo pixel outside the canvas is conv2's zero padding, not conv1(0): only tiles on the canvas border have any
            const bool interior = oy0 >= 2 && ox0 >= 2 && oy0 + TH + 2 <= a.Hin && ox0 + TW + 2 <= a.Win;   // wave-uniform
            bool colT[4], incT[4];
#pragma unroll
            for (int ty = 0; ty < 4; ++ty) {
                const int gx = ox0 - 2 + hxT[ty];
                colT[ty] = gx >= 0 && gx < a.Win;
                const int gy = oy0 - 2 + 5 * pw + jT[ty];
                incT[ty] = colT[ty] && gy >= 0 && gy < a.Hin;
            }
            // straight-line, three batches of four tiles: all fragment reads of a batch, its 16 MFMAs (the two k-steps of a
            // tile are eight MFMAs apart), then the four epilogues -- no branches inside, so the scheduler overlaps the batches
            auto batches = [&](auto border_tag) {
                constexpr bool BORDER = decltype(border_tag)::value;
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    uint4 xa[4], xb[4];
                    int d1[4], d2[4];
                    bool inc[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int k = b * 4 + q;                       // tiles 0-4: ct 0 rows 0-4, 5-9: ct 1, 10: type 2, 11: type 3
                        const int ty = k < 5 ? 0 : (k < 10 ? 1 : k - 8);
                        const int j = k < 10 ? k % 5 : 0;              // immediate row step of the full tiles
                        const uint32_t* ra = (const uint32_t*)(priv + srcA[ty] + j * UCB);
                        const uint32_t* rb = (const uint32_t*)(priv + srcB[ty] + j * UCB);
                        xa[q] = make_uint4(ra[0], ra[1], ra[2], ra[3]);
                        xb[q] = make_uint4(rb[0], rb[1], rb[2], rb[3]);
                        d1[q] = bufoff + dst1[ty] + j * ROWP;
                        // couts 16-19 (lane group 0) [and the zero pad 20-23 (lane group 1) without PAIRC2]
                        d2[q] = (PAIRC2 ? g == 0 : low_g) ? bufoff + dst2[ty] + j * ROWP : sink;
                        inc[q] = true;
                        if constexpr (BORDER) {
                            if (k < 10) { const int gy = oy0 - 2 + 5 * pw + j; inc[q] = colT[ty] && gy >= 0 && gy < a.Hin; }
                            else inc[q] = incT[ty];
                        }
                    }
                    f32x4 z0[4], z1[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        z0[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[0][0], __builtin_bit_cast(bf16x8, xa[q]), zb0, 0, 0, 0);
                        z1[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[0][1], __builtin_bit_cast(bf16x8, xa[q]), zb1, 0, 0, 0);
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        z0[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[1][0], __builtin_bit_cast(bf16x8, xb[q]), z0[q], 0, 0, 0);
                        z1[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[1][1], __builtin_bit_cast(bf16x8, xb[q]), z1[q], 0, 0, 0);
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        uint32_t pa0 = relu_pk_bf16(pk_bf16(z0[q][0], z0[q][1]), floor2), pa1 = relu_pk_bf16(pk_bf16(z0[q][2], z0[q][3]), floor2);
                        uint32_t pb0 = relu_pk_bf16(pk_bf16(z1[q][0], z1[q][1]), floor2), pb1 = relu_pk_bf16(pk_bf16(z1[q][2], z1[q][3]), floor2);
                        if constexpr (BORDER) {
                            pa0 = inc[q] ? pa0 : 0u; pa1 = inc[q] ? pa1 : 0u;
                            pb0 = inc[q] ? pb0 : 0u; pb1 = inc[q] ? pb1 : 0u;
                        }
                        *(uint2*)(smem + d1[q]) = make_uint2(pa0, pa1);
                        *(uint2*)(smem + d2[q]) = make_uint2(pb0, pb1);
                        if constexpr (PAIRC2)       // ... and once more as the left neighbour's right half (its slot ends 8 bytes before ours)
                            *(uint2*)(smem + (g == 0 ? d2[q] - 40 : sink)) = make_uint2(pb0, pb1);
                    }
                }
            };
            // The same first layer on v_mfma_f32_32x32x16_bf16 (default): a 32-pixel tile needs 3 MFMAs of 32 cycles (kernel rows
            // 2 s + half per k-step) instead of 2 x 4 of 16, three fragment reads instead of four, and its 32 x 32 result holds a
            // pixel's 24 channel slots in twelve registers of the two lane halves -- six roundings per lane instead of eight.
            // The wave's band = five full tiles (row j, halo columns 0-31) + one tile of the four remainder columns of all rows.
            auto batches32 = [&](auto border_tag) {
                constexpr bool BORDER = decltype(border_tag)::value;
                typedef float f32x16 __attribute__((ext_vector_type(16)));
                // this fill's store addresses: the per-kernel lane constants + the tile buffer's offset (rows ride in immediates)
                int d12[2], d3[2], d4[2];
#pragma unroll
                for (int ty = 0; ty < 2; ++ty) { d12[ty] = c32_d12[ty] + bufoff; d3[ty] = c32_d3[ty] + bufoff; d4[ty] = c32_d4[ty] + bufoff; }
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    uint4 x3[3][3];
                    bool inc[3];
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        const int k = b * 3 + q;                           // tiles 0-4: full, row k; 5: remainder columns
                        const int ty = k == 5 ? 1 : 0, jimm = k == 5 ? 0 : k;
#pragma unroll
                        for (int s3 = 0; s3 < 3; ++s3) {
                            const uint32_t* r = (const uint32_t*)(priv + c32_src[ty][s3] + jimm * UCB);
                            x3[q][s3] = make_uint4(r[0], r[1], r[2], r[3]);
                        }
                        inc[q] = true;
                        if constexpr (BORDER) {
                            const int gx = ox0 - 2 + c32_hx[ty], gy = oy0 - 2 + 5 * pw + c32_j[ty] + jimm;
                            inc[q] = gx >= 0 && gx < a.Win && gy >= 0 && gy < a.Hin;
                        }
                    }
                    f32x16 z[3];
#pragma unroll
                    for (int q = 0; q < 3; ++q) z[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w3[0], __builtin_bit_cast(bf16x8, x3[q][0]), zb32, 0, 0, 0);
#pragma unroll
                    for (int s3 = 1; s3 < 3; ++s3)
#pragma unroll
                        for (int q = 0; q < 3; ++q) z[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w3[s3], __builtin_bit_cast(bf16x8, x3[q][s3]), z[q], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        const int k = b * 3 + q, ty = k == 5 ? 1 : 0, ro = (k == 5 ? 0 : k) * ROWP;
                        // registers 4 v .. 4 v + 3 = couts 8 v + 4 half .. + 3: bytes 16 v + 8 half of the pixel's slot
                        uint32_t pk[3][2];
#pragma unroll
                        for (int v = 0; v < 3; ++v) {
                            pk[v][0] = relu_pk_bf16(pk_bf16(z[q][4 * v], z[q][4 * v + 1]), floor2);
                            pk[v][1] = relu_pk_bf16(pk_bf16(z[q][4 * v + 2], z[q][4 * v + 3]), floor2);
                            if constexpr (BORDER) { pk[v][0] = inc[q] ? pk[v][0] : 0u; pk[v][1] = inc[q] ? pk[v][1] : 0u; }
                        }
                        *(uint2*)(smem + d12[ty] + ro) = make_uint2(pk[0][0], pk[0][1]);               // couts 0-3 / 4-7
                        *(uint2*)(smem + d12[ty] + ro + 16) = make_uint2(pk[1][0], pk[1][1]);          // couts 8-11 / 12-15
                        // couts 16-19 (lane half 0): the own slot [PAIRC2: and the left neighbour's right half]; lane half 1 holds
                        // the zero pad 20-23 [PAIRC2: nothing -- its stores land in the row's 16 padding bytes]
                        *(uint2*)(smem + d3[ty] + ro) = make_uint2(pk[2][0], pk[2][1]);
                        if constexpr (PAIRC2) *(uint2*)(smem + d4[ty] + ro) = make_uint2(pk[2][0], pk[2][1]);
                    }
                }
            };
            if (a.f1_wpk32) {
                if (interior) batches32(std::false_type{});
                else batches32(std::true_type{});
            } else {
                if (interior) batches(std::false_type{});
                else batches(std::true_type{});
            }
        };
        tile_origin(0, noy, nox);
        load_u8(noy, nox);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // weight DMA pieces of this wave + the first page bytes
        lds_barrier();                                         // (A) weights and table resident
        fill(0, 0);
        lds_barrier();                                         // (B) tile 0 is full
        long long tf = 0, tw = 0;
        for (int i = 0; i < n_my; ++i) {
            const long long c0 = a.trace ? (long long)__builtin_amdgcn_s_memtime() : 0;
            if (i + 1 < n_my) fill(i + 1, ((i + 1) & 1) * TB);
            const long long c1 = a.trace ? (long long)__builtin_amdgcn_s_memtime() : 0;
            lds_barrier();                                     // tile i + 1 full, tile i drained
            if (a.trace) { tf += c1 - c0; tw += (long long)__builtin_amdgcn_s_memtime() - c1; }
        }
        if (a.trace && lane == 0) {
            unsigned long long* o = a.trace + ((size_t)blockIdx.x * 8 + wave) * 4;
            o[0] = (unsigned long long)tf; o[1] = 0; o[2] = (unsigned long long)tw; o[3] = (unsigned long long)n_my;
        }
        return;
    }

    // =============================== CONSUMERS ===============================
    // A wave owns four rows x 32 pixels of the tile = eight 16-pixel tiles m (row m >> 1, column tile m & 1) x two cout tiles.
    // Two forms of the tile loop (bit-identical results): the default runs ONE k-loop over all eight pixel tiles and then the
    // epilogue; PSEG_WS_FORM=1 computes the tile in two phases of four pixel tiles (rows 0-1, then rows 2-3) and issues the
    // epilogue of the half finished just before -- fused 2x2 max-pool (its row pairs lie inside a half), bf16 rounding, the
    // skip-logits MFMAs, all stores -- in pieces between the current phase's MFMAs.
    // B fragment of pixel tile m for k-step s: one ds_read_b128 at (tile + lane pixel + chunk offset of (s, g)) + an
    // immediate ((m >> 1) rows, (m & 1) * 16 pixels); A fragment (s, t): weights + lane * 16 + an immediate.  The 19 chunk
    // offsets of this lane group stay in registers for the whole kernel.
    if ((a.dbg & 0x300) == 0x200) __builtin_amdgcn_s_setprio(1);       // PSEG_WS_PRIO=2: the consumers win issue arbitration
    const float4 bias0 = *(const float4*)(a.bias + 4 * g), bias1 = *(const float4*)(a.bias + 16 + 4 * g);
    bf16x8 wq;
    if constexpr (SKIPLOG) wq = *(const bf16x8*)(a.tail_wa + lane * 8);
    const int CsO = a.nch_out * 8;
    constexpr unsigned OOBS = 0xfffffff0u;
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)a.dst, 0, a.dst_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)a.pool_dst, 0, a.pool_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(SKIPLOG ? (void*)a.skip_logits : (void*)a.dst), 0,
                                                                        SKIPLOG ? (unsigned)((size_t)a.Hout * a.Wout * a.skip_CP * 4) : 0u, 0x00020000);
    const int Wo2 = a.Wout >> 1, Ho2 = a.Hout >> 1;
    const unsigned noff0 = 4 * g < CsO ? (unsigned)(4 * g) * 2u : OOBS, noff1 = 16 + 4 * g < CsO ? (unsigned)(16 + 4 * g) * 2u : OOBS;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // weight DMA pieces of this wave
    lds_barrier();                                             // (A)
    int offv[KSTEPS];
#pragma unroll
    for (int s2 = 0; s2 < KSTEPS; ++s2) offv[s2] = tab_l[s2 * 4 + g] + (wave * (MT / 2)) * ROWP + p16 * PS2;
    const char* const wb = w_t + lane * 16;
    lds_barrier();                                             // (B)

    f32x4 accA[4][NT], accB[4][NT];                            // the two halves: pixel tiles 0-3 (rows 0, 1) and 4-7 (rows 2, 3)
    uint2 pks[NT][4];                                          // bf16 pairs of the half being drained (B operand of the skip-logits MFMA)
    // one piece of the epilogue of half `HB` (0: rows 0-1, 1: rows 2-3) of the tile at (doy, dox); `acc` holds it
    // P(t2, c): column tile c (local tiles c and c + 2 are vertical neighbours): rounding + pool + stores of cout tile t2
    auto piece_P = [&](f32x4 (&acc)[4][NT], int hb, int doy, int dox, bool valid, auto t2c, auto cc) {
        constexpr int t2 = decltype(t2c)::value, c = decltype(cc)::value;
        const unsigned noff = t2 == 0 ? noff0 : noff1;
        float q[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) q[r] = vmax_xor1(vmax(acc[c][t2][r], acc[c + 2][t2][r]));
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x4 v = acc[c + 2 * h][t2];
            const uint2 pk = make_uint2(pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3]));
            if constexpr (SKIPLOG) { if constexpr (FORM != 2) pks[t2][c + 2 * h] = pk; }
            else {
                const int y = doy + wave * (MT / 2) + 2 * hb + h, x = dox + c * 16 + p16;
                const unsigned o = (valid && y < a.Hout && x < a.Wout && noff != OOBS) ? (unsigned)(y * a.Wout + x) * (unsigned)(CsO * 2) + noff : OOBS;
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, pk), rd, o, 0, 0);
            }
        }
        const int y = (doy >> 1) + wave * (MT / 4) + hb;
        const int x = (dox >> 1) + ((c * 16 + p16) >> 1);
        const bool ok = valid && !(p16 & 1) && y < Ho2 && x < Wo2 && noff != OOBS;
        const unsigned o = ok ? (unsigned)(y * Wo2 + x) * (unsigned)(CsO * 2) + noff : OOBS;
        const uint2 pk = make_uint2(pk_bf16(q[0], q[1]), pk_bf16(q[2], q[3]));
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, pk), rp, o, 0, 0);
    };
    // S(c): the skip-logits products of local tiles c and c + 2 (both cout tiles rounded by then)
    auto piece_S = [&](int hb, int doy, int dox, bool valid, auto cc, f32x4 (*sacc)[NT] = nullptr) {
        constexpr int c = decltype(cc)::value;
        if constexpr (SKIPLOG) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int lm = c + 2 * h;
                f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
                uint4 bq;
                if constexpr (FORM == 2) {   // no registers for the pairs piece_P rounded: round them again from the accumulators
                    const f32x4 v0 = sacc[lm][0], v1 = sacc[lm][1];
                    bq = make_uint4(pk_bf16(v0[0], v0[1]), pk_bf16(v0[2], v0[3]), pk_bf16(v1[0], v1[1]), pk_bf16(v1[2], v1[3]));
                } else {
                    bq = make_uint4(pks[0][lm].x, pks[0][lm].y, pks[1][lm].x, pks[1][lm].y);
                }
                z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq, __builtin_bit_cast(bf16x8, bq), z, 0, 0, 0);
                const int y = doy + wave * (MT / 2) + 2 * hb + h, x = dox + c * 16 + p16;
                const unsigned o = (valid && y < a.Hout && x < a.Wout && 4 * g < a.skip_CP) ? ((unsigned)(y * a.Wout + x) * (unsigned)a.skip_CP + 4u * g) * 4u : OOBS;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, z), rs, o, 0, 0);
            }
        }
    };
    // one phase: accumulate half `hb` of the tile in `in_t` into `acc`; between the MFMAs, drain `dacc` = half `dhb` of the tile
    // at (doy, dox) (valid = such a half exists)
    auto phase = [&](const char* in_t, f32x4 (&acc)[4][NT], auto hbc, f32x4 (&dacc)[4][NT], int dhb, int doy, int dox, bool valid) {
        constexpr int hb = decltype(hbc)::value;
#pragma unroll
        for (int m = 0; m < 4; ++m) {                           // start value = bias (the first MFMA's C operand: no add later)
            acc[m][0] = f32x4{bias0.x, bias0.y, bias0.z, bias0.w};
            acc[m][1] = f32x4{bias1.x, bias1.y, bias1.z, bias1.w};
        }
        bf16x8 xs[2][4], ws2[2][NT];
#define WS_LOAD(SET, S)                                                                                   \
        {                                                                                                 \
            const char* va_ = in_t + offv[S];                                                             \
            ws2[SET][0] = *(const bf16x8*)(wb + ((S) * NT) * 1024);                                       \
            _Pragma("unroll") for (int m = 0; m < 4; ++m)                                                 \
                xs[SET][m] = *(const bf16x8*)(va_ + (2 * hb + (m >> 1)) * ROWP + (m & 1) * 16 * PS2);     \
            ws2[SET][1] = *(const bf16x8*)(wb + ((S) * NT + 1) * 1024);                                   \
        }
        WS_LOAD(0, 0)
#pragma unroll
        for (int s2 = 0; s2 < KSTEPS; ++s2) {
            __builtin_amdgcn_sched_barrier(0);                  // a k-step is one scheduling region
            if (s2 + 1 < KSTEPS) WS_LOAD((s2 + 1) & 1, s2 + 1)
#pragma unroll
            for (int t2 = 0; t2 < NT; ++t2)
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    acc[m][t2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ws2[s2 & 1][t2], xs[s2 & 1][m], acc[m][t2], 0, 0, 0);
            // the drained half's epilogue, one piece per k-step (P needs nothing; S(c) needs P(0, c) and P(1, c))
            if (s2 == 1) piece_P(dacc, dhb, doy, dox, valid, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
            if (s2 == 3) piece_P(dacc, dhb, doy, dox, valid, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
            if (s2 == 5) piece_S(dhb, doy, dox, valid, std::integral_constant<int, 0>{});
            if (s2 == 7) piece_P(dacc, dhb, doy, dox, valid, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
            if (s2 == 9) piece_P(dacc, dhb, doy, dox, valid, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
            if (s2 == 11) piece_S(dhb, doy, dox, valid, std::integral_constant<int, 1>{});
            // issue order: the address add, then behind each MFMA one fragment read of the next step (six of them) and two
            // VALU instructions of the epilogue piece
            if (s2 + 1 < KSTEPS) __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (r < 6 && s2 + 1 < KSTEPS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, PSEG_WS_EPI_VALU, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#undef WS_LOAD
    };
    long long tk = 0, te = 0, tw = 0;
    int poy = 0, pox = 0;
    if constexpr (FORM == 2) {
        // One k-loop per tile (every A fragment serves eight pixel tiles) AND no epilogue phase: the finished tile's accumulators
        // stay in a second register set and drain in twelve pieces between the MFMAs of the next tile's k-loop.
        f32x4 accC[4][NT], accD[4][NT];                         // second accumulator set (halves as accA / accB)
        auto tile_pass = [&](const char* in_t, f32x4 (&cA)[4][NT], f32x4 (&cB)[4][NT], f32x4 (&dA)[4][NT], f32x4 (&dB)[4][NT],
                             int doy, int dox, bool valid) {
            {   // (the bias is re-read per tile: eight registers that would otherwise live through the k-loop)
                const float4 bb0 = *(const float4*)(a.bias + 4 * g), bb1 = *(const float4*)(a.bias + 16 + 4 * g);
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    cA[m][0] = cB[m][0] = f32x4{bb0.x, bb0.y, bb0.z, bb0.w};
                    cA[m][1] = cB[m][1] = f32x4{bb1.x, bb1.y, bb1.z, bb1.w};
                }
            }
            // Registers are the limit of this form (two accumulator sets): ONE set of pixel fragments -- the MFMAs of a k-step run
            // pixel tile by pixel tile (both cout tiles of tile m back to back), and the next step's fragment of tile m is read
            // into the same registers right behind them, 14 MFMAs before its first use -- two sets of kernel fragments, and the 17
            // chunk offsets come from the LDS table one k-step ahead instead of living in registers.
            bf16x8 xs[MT], ws2[2][NT];
            const char* const bt = in_t + (wave * (MT / 2)) * ROWP + p16 * PS2;
            const int* const tbl = tab_l + g;
            int toff[2];
            toff[0] = tbl[0];
            toff[1] = tbl[4];
            {
                const char* va_ = bt + toff[0];
                ws2[0][0] = *(const bf16x8*)(wb);
#pragma unroll
                for (int m = 0; m < MT; ++m) xs[m] = *(const bf16x8*)(va_ + (m >> 1) * ROWP + (m & 1) * 16 * PS2);
                ws2[0][1] = *(const bf16x8*)(wb + 1024);
            }
#pragma unroll
            for (int s2 = 0; s2 < KSTEPS; ++s2) {
                __builtin_amdgcn_sched_barrier(0);
                const bool more = s2 + 1 < KSTEPS;
                const char* va_ = bt + toff[(s2 + 1) & 1];
                if (more) {
                    ws2[(s2 + 1) & 1][0] = *(const bf16x8*)(wb + ((s2 + 1) * NT) * 1024);
                    ws2[(s2 + 1) & 1][1] = *(const bf16x8*)(wb + ((s2 + 1) * NT + 1) * 1024);
                }
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    f32x4& a0_ = m < 4 ? cA[m][0] : cB[m - 4][0];
                    f32x4& a1_ = m < 4 ? cA[m][1] : cB[m - 4][1];
                    a0_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ws2[s2 & 1][0], xs[m], a0_, 0, 0, 0);
                    a1_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ws2[s2 & 1][1], xs[m], a1_, 0, 0, 0);
                    if (more) xs[m] = *(const bf16x8*)(va_ + (m >> 1) * ROWP + (m & 1) * 16 * PS2);
                }
                if (s2 + 2 < KSTEPS) toff[s2 & 1] = tbl[(s2 + 2) * 4];        // slot s2 & 1 was consumed by the previous region's reads
                // the previous tile's epilogue: half 0 in k-steps 1-6, half 1 in k-steps 7-12
                if (s2 >= 1 && s2 <= 12) {
                    const int hb = s2 <= 6 ? 0 : 1, pc = (s2 - 1) % 6;
                    f32x4 (&hacc)[4][NT] = s2 <= 6 ? dA : dB;
                    if (pc == 0) piece_P(hacc, hb, doy, dox, valid, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
                    if (pc == 1) piece_P(hacc, hb, doy, dox, valid, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
                    if (pc == 2) piece_S(hb, doy, dox, valid, std::integral_constant<int, 0>{}, hacc);
                    if (pc == 3) piece_P(hacc, hb, doy, dox, valid, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
                    if (pc == 4) piece_P(hacc, hb, doy, dox, valid, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
                    if (pc == 5) piece_S(hb, doy, dox, valid, std::integral_constant<int, 1>{}, hacc);
                }
                // issue order: the address add and the two kernel-fragment reads, then per pixel tile its two MFMAs, the read of its
                // next fragment and two epilogue VALU instructions
                if (more) {
                    __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
#pragma unroll
                for (int r = 0; r < MT; ++r) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    if (more) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        // two tiles per trip: the accumulator sets keep their roles (A/B accumulate even tiles, C/D odd ones), so no register
        // moves at the loop edge
        auto origin = [&](int i, int& oy, int& ox) {
            const int t = xcd_tile((int)blockIdx.x + i * (int)gridDim.x);
            const int ty = t / tiles_x, tx = t - ty * tiles_x;
            oy = ty * TH; ox = tx * TW;
        };
        for (int i = 0; i < n_my; i += 2) {
            long long c0 = a.trace ? (long long)__builtin_amdgcn_s_memtime() : 0;
            int oy0, ox0;
            origin(i, oy0, ox0);
            tile_pass(smem, accA, accB, accC, accD, poy, pox, i > 0);              // even tile: buffer 0
            poy = oy0; pox = ox0;
            long long c2 = a.trace ? (long long)__builtin_amdgcn_s_memtime() : 0;
            lds_barrier();
            if (a.trace) { tk += c2 - c0; tw += (long long)__builtin_amdgcn_s_memtime() - c2; }
            if (i + 1 < n_my) {
                c0 = a.trace ? (long long)__builtin_amdgcn_s_memtime() : 0;
                origin(i + 1, oy0, ox0);
                tile_pass(smem + TB, accC, accD, accA, accB, poy, pox, true);      // odd tile: buffer 1
                poy = oy0; pox = ox0;
                c2 = a.trace ? (long long)__builtin_amdgcn_s_memtime() : 0;
                lds_barrier();
                if (a.trace) { tk += c2 - c0; tw += (long long)__builtin_amdgcn_s_memtime() - c2; }
            }
        }
        {   // the last tile drains on its own
            const long long c1 = a.trace ? (long long)__builtin_amdgcn_s_memtime() : 0;
            asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
            const bool odd = (n_my & 1) == 0;                     // the last tile (n_my - 1) used set C/D when its index is odd
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                if (odd) {
                    f32x4 (&hacc)[4][NT] = hb == 0 ? accC : accD;
                    piece_P(hacc, hb, poy, pox, n_my > 0, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
                    piece_P(hacc, hb, poy, pox, n_my > 0, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
                    piece_S(hb, poy, pox, n_my > 0, std::integral_constant<int, 0>{}, hacc);
                    piece_P(hacc, hb, poy, pox, n_my > 0, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
                    piece_P(hacc, hb, poy, pox, n_my > 0, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
                    piece_S(hb, poy, pox, n_my > 0, std::integral_constant<int, 1>{}, hacc);
                } else {
                    f32x4 (&hacc)[4][NT] = hb == 0 ? accA : accB;
                    piece_P(hacc, hb, poy, pox, n_my > 0, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
                    piece_P(hacc, hb, poy, pox, n_my > 0, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
                    piece_S(hb, poy, pox, n_my > 0, std::integral_constant<int, 0>{}, hacc);
                    piece_P(hacc, hb, poy, pox, n_my > 0, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
                    piece_P(hacc, hb, poy, pox, n_my > 0, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
                    piece_S(hb, poy, pox, n_my > 0, std::integral_constant<int, 1>{}, hacc);
                }
            }
            if (a.trace) te += (long long)__builtin_amdgcn_s_memtime() - c1;
        }
        if (a.trace && lane == 0) {
            unsigned long long* o = a.trace + ((size_t)blockIdx.x * 8 + wave) * 4;
            o[0] = (unsigned long long)tk; o[1] = (unsigned long long)te; o[2] = (unsigned long long)tw; o[3] = (unsigned long long)n_my;
        }
        return;
    }
    if constexpr (FORM == 0) {
        // Default form: the whole tile in ONE k-loop (eight pixel tiles per A fragment) and the epilogue after it; the producer
        // on the same SIMD gets the matrix pipe and most issue slots meanwhile.  The two-phase form below (PSEG_WS_FORM=1) hides
        // the epilogue behind the consumer's own MFMAs but reads every A fragment twice (+17 % LDS reads): 99.5 vs 94.5 us.
        for (int i = 0; i < n_my; ++i) {
            const long long c0 = a.trace ? (long long)__builtin_amdgcn_s_memtime() : 0;
            const char* in_t = smem + (i & 1) * TB;
            const int t = xcd_tile((int)blockIdx.x + i * (int)gridDim.x);
            const int ty = t / tiles_x, tx = t - ty * tiles_x;
            const int oy0 = ty * TH, ox0 = tx * TW;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                accA[m][0] = accB[m][0] = f32x4{bias0.x, bias0.y, bias0.z, bias0.w};
                accA[m][1] = accB[m][1] = f32x4{bias1.x, bias1.y, bias1.z, bias1.w};
            }
            {
                bf16x8 xs[2][MT], ws2[2][NT];
#define WS_LOAD8(SET, S)                                                                              \
                {                                                                                     \
                    const char* va_ = in_t + offv[S];                                                 \
                    ws2[SET][0] = *(const bf16x8*)(wb + ((S) * NT) * 1024);                           \
                    _Pragma("unroll") for (int m = 0; m < MT; ++m)                                    \
                        xs[SET][m] = *(const bf16x8*)(va_ + (m >> 1) * ROWP + (m & 1) * 16 * PS2);    \
                    ws2[SET][1] = *(const bf16x8*)(wb + ((S) * NT + 1) * 1024);                       \
                }
                WS_LOAD8(0, 0)
#pragma unroll
                for (int s2 = 0; s2 < KSTEPS; ++s2) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (s2 + 1 < KSTEPS) WS_LOAD8((s2 + 1) & 1, s2 + 1)
#pragma unroll
                    for (int t2 = 0; t2 < NT; ++t2)
#pragma unroll
                        for (int m = 0; m < MT; ++m) {
                            f32x4& ac = m < 4 ? accA[m][t2] : accB[m - 4][t2];
                            ac = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ws2[s2 & 1][t2], xs[s2 & 1][m], ac, 0, 0, 0);
                        }
                    if (s2 + 1 < KSTEPS) {
                        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
#pragma unroll
                        for (int r = 0; r < 10; ++r) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        }
                        __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#undef WS_LOAD8
            }
            const long long c1 = a.trace ? (long long)__builtin_amdgcn_s_memtime() : 0;
            asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                f32x4 (&hacc)[4][NT] = hb == 0 ? accA : accB;
                piece_P(hacc, hb, oy0, ox0, true, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
                piece_P(hacc, hb, oy0, ox0, true, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
                piece_S(hb, oy0, ox0, true, std::integral_constant<int, 0>{});
                piece_P(hacc, hb, oy0, ox0, true, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
                piece_P(hacc, hb, oy0, ox0, true, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
                piece_S(hb, oy0, ox0, true, std::integral_constant<int, 1>{});
            }
            const long long c2 = a.trace ? (long long)__builtin_amdgcn_s_memtime() : 0;
            lds_barrier();
            if (a.trace) { tk += c1 - c0; te += c2 - c1; tw += (long long)__builtin_amdgcn_s_memtime() - c2; }
        }
        if (a.trace && lane == 0) {
            unsigned long long* o = a.trace + ((size_t)blockIdx.x * 8 + wave) * 4;
            o[0] = (unsigned long long)tk; o[1] = (unsigned long long)te; o[2] = (unsigned long long)tw; o[3] = (unsigned long long)n_my;
        }
        return;
    }
    for (int i = 0; i < n_my; ++i) {
        const long long c0 = a.trace ? (long long)__builtin_amdgcn_s_memtime() : 0;
        const char* in_t = smem + (i & 1) * TB;
        const int t = xcd_tile((int)blockIdx.x + i * (int)gridDim.x);
        const int ty = t / tiles_x, tx = t - ty * tiles_x;
        const int oy0 = ty * TH, ox0 = tx * TW;
        phase(in_t, accA, std::integral_constant<int, 0>{}, accB, 1, poy, pox, i > 0);     // rows 0-1; drains rows 2-3 of the previous tile
        phase(in_t, accB, std::integral_constant<int, 1>{}, accA, 0, oy0, ox0, true);      // rows 2-3; drains rows 0-1 of this tile
        poy = oy0; pox = ox0;
        const long long c2 = a.trace ? (long long)__builtin_amdgcn_s_memtime() : 0;
        lds_barrier();                                         // tile i drained, tile i + 1 full
        if (a.trace) { tk += c2 - c0; tw += (long long)__builtin_amdgcn_s_memtime() - c2; }
    }
    {   // rows 2-3 of the last tile: nothing left to hide behind
        const long long c1 = a.trace ? (long long)__builtin_amdgcn_s_memtime() : 0;
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");   // (inline-asm readers of accumulator registers follow)
        piece_P(accB, 1, poy, pox, n_my > 0, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        piece_P(accB, 1, poy, pox, n_my > 0, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
        piece_S(1, poy, pox, n_my > 0, std::integral_constant<int, 0>{});
        piece_P(accB, 1, poy, pox, n_my > 0, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
        piece_P(accB, 1, poy, pox, n_my > 0, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
        piece_S(1, poy, pox, n_my > 0, std::integral_constant<int, 1>{});
        if (a.trace) te += (long long)__builtin_amdgcn_s_memtime() - c1;
    }
    if (a.trace && lane == 0) {
        unsigned long long* o = a.trace + ((size_t)blockIdx.x * 8 + wave) * 4;
        o[0] = (unsigned long long)tk; o[1] = (unsigned long long)te; o[2] = (unsigned long long)tw; o[3] = (unsigned long long)n_my;
    }
}

// ---------------------------------------------------------------------------------------------
// Composed tail (fcn / fcn_skip): Conv2DTranspose k2 s2 (linear) -> [concat skip] -> crop ->
// logits 1x1 -> softmax / argmax as ONE small GEMM per half-resolution pixel.  deconv5 has no
// activation, so deconv5 o logits is linear in deconv5's input x:
//     logit[ab][c] = sum_ci M_ab[c][ci] x[ci] + sum_cs Wl_skip[c][cs] skip_ab[cs] + beta[c],
//     M_ab = Wl_dec . Wd[ab],   beta = bl + Wl_dec . bd            (host, from the bf16 kernels)
// Rows of the MFMA tile are (sub-pixel ab, class): row = ab*CP + c with CP = 4 / 8 / 16 classes
// padded, so one 16x16x32 MFMA per k-step yields all four output pixels of a half-res pixel for
// C <= 4 (lane group g = sub-pixel, 4 registers = classes: the argmax never leaves the lane).
// The skip tensor enters through zero-extended A operands (rows of the other sub-pixels are 0).
// No LDS, no spatial reuse: every B fragment is one coalesced 16-byte global load per lane; HBM-
// bound (reads x and skip exactly once: 113 + 201 MB at 2048x1536, writes the label map).
// ---------------------------------------------------------------------------------------------
struct TailC {
    const uint16_t* src0; const uint16_t* src1; const uint16_t* skip;
    int nch0, nch1, nch_skip;      // 16-byte chunks per pixel
    int Hh, Wh;                    // half-resolution canvas
    int H0, W0, C;                 // output (cropped) size, classes
    const uint16_t* wA1;           // [tile][k-step][64][8]
    const uint16_t* wA2;           // [ab][64][8]
    const float* beta;             // [NTL*16]
    const float* S;                // skip-logits buffer [full-res canvas pixel][CP] written by the skip producer, or null
    float* out_logits; float* out_probs; int64_t* out_labels; uint8_t* out_labels_u8;
    float* out_margin;             // top-1 minus top-2 logit per pixel, or null
};

template <int CP, int NKS0, int NKSS>
__global__ __launch_bounds__(256) void tail_composed_kernel(TailC a) {
    constexpr int NTL = CP / 4;          // 16-row tiles: rows = ab*CP + class
    const int lane = threadIdx.x & 63, p16 = lane & 15, g = lane >> 4;
    const int wv = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int tiles_x = a.Wh >> 4;
    const int hy = wv / tiles_x, hx = (wv - hy * tiles_x) * 16 + p16;
    if (hy >= a.Hh) return;
    // ---- all loads first ----
    uint4 xb[NKS0];
#pragma unroll
    for (int s = 0; s < NKS0; ++s) {
        const int ch = 4 * s + g;
        xb[s] = make_uint4(0, 0, 0, 0);
        const size_t px = (size_t)hy * a.Wh + hx;
        if (ch < a.nch0) xb[s] = *(const uint4*)(a.src0 + (px * a.nch0 + ch) * 8);
        else if (ch < a.nch0 + a.nch1) xb[s] = *(const uint4*)(a.src1 + (px * a.nch1 + (ch - a.nch0)) * 8);
    }
    uint4 sb[4][NKSS > 0 ? NKSS : 1];
    if constexpr (NKSS > 0) {
#pragma unroll
        for (int ab = 0; ab < 4; ++ab)
#pragma unroll
            for (int s = 0; s < NKSS; ++s) {
                const int ch = 4 * s + g;
                const size_t px = (size_t)(2 * hy + (ab >> 1)) * (2 * a.Wh) + 2 * hx + (ab & 1);
                sb[ab][s] = ch < a.nch_skip ? *(const uint4*)(a.skip + (px * a.nch_skip + ch) * 8) : make_uint4(0, 0, 0, 0);
            }
    }
    f32x4 acc[NTL];
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
        const float4 b = *(const float4*)(a.beta + t * 16 + 4 * g);
        acc[t] = f32x4{b.x, b.y, b.z, b.w};
    }
#pragma unroll
    for (int s = 0; s < NKS0; ++s)
#pragma unroll
        for (int t = 0; t < NTL; ++t) {
            const bf16x8 w = *(const bf16x8*)(a.wA1 + ((size_t)(t * NKS0 + s) * 64 + lane) * 8);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, __builtin_bit_cast(bf16x8, xb[s]), acc[t], 0, 0, 0);
        }
    if constexpr (NKSS > 0) {
#pragma unroll
        for (int ab = 0; ab < 4; ++ab)
#pragma unroll
            for (int s = 0; s < NKSS; ++s) {
                const bf16x8 w = *(const bf16x8*)(a.wA2 + ((size_t)(ab * NKSS + s) * 64 + lane) * 8);
                constexpr int dummy = 0; (void)dummy;
                const int t = ab * CP / 16;
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, __builtin_bit_cast(bf16x8, sb[ab][s]), acc[t], 0, 0, 0);
            }
    }
    if (a.S) {   // skip contribution precomputed by the producer of the skip tensor: this lane's four classes of its sub-pixel
#pragma unroll
        for (int t = 0; t < NTL; ++t) {
            const int row0 = 16 * t + 4 * g, ab = row0 / CP, c0 = row0 - ab * CP;
            const size_t px = (size_t)(2 * hy + (ab >> 1)) * (2 * a.Wh) + 2 * hx + (ab & 1);
            const float4 sv = *(const float4*)(a.S + px * CP + c0);
            acc[t][0] += sv.x; acc[t][1] += sv.y; acc[t][2] += sv.z; acc[t][3] += sv.w;
        }
    }
    // ---- per pixel: classes live in CP/4 lane groups (xor 16 / 32 partners) ----
    const int C = a.C;
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
        const int row0 = 16 * t + 4 * g;
        const int ab = row0 / CP, c0 = row0 - ab * CP;
        const int y = 2 * hy + (ab >> 1), x = 2 * hx + (ab & 1);
        const bool inb = y < a.H0 && x < a.W0;
        float bv, sv;
        int bi;
        top2_classes<CP / 4>(acc[t], c0, C, bv, bi, sv);
        const size_t p = (size_t)y * a.W0 + x;
        if (inb && c0 == 0) {
            if (a.out_labels_u8) a.out_labels_u8[p] = (uint8_t)bi;
            if (a.out_labels) a.out_labels[p] = bi;
            if (a.out_margin) a.out_margin[p] = bv - sv;
        }
        if (a.out_logits && inb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (c0 + r < C) a.out_logits[p * C + c0 + r] = acc[t][r];
        if (a.out_probs) {
            float ex[4], sum = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) { ex[r] = (c0 + r < C) ? expf(acc[t][r] - bv) : 0.f; sum += ex[r]; }
            if constexpr (CP > 4) {
#pragma unroll
                for (int sh = 16; sh < 4 * CP; sh <<= 1) sum += __shfl_xor(sum, sh);
            }
            if (inb)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (c0 + r < C) a.out_probs[p * C + c0 + r] = ex[r] / sum;
        }
    }
}


// ---------------------------------------------------------------------------------------------
// Composed tail with the preceding transposed conv inside (fcn_skip: deconv4 -> [concat conv3] -> deconv5 o logits).
// deconv4 (k2 s2, ReLU) maps a quarter-resolution pixel q to four half-resolution pixels; its 30-channel output is
// read by nothing but this tail, so it is recomputed here and never stored: a wave takes 16 half-resolution pixels
// of ONE sub-pixel parity (hx = x0 + 2 p16 + b, a = hy & 1), so that they share one weight matrix Wd4[ab] and the
// GEMM d4[co][pixel] = Wd4[ab] . [deconv3 ; conv5](q) is four k-steps of two MFMAs.  +bias, ReLU and the bf16
// rounding happen in registers; the two accumulator tiles are then the B operand of the composed deconv5 o logits
// GEMM (k = 8g+j <-> channel 4g+j / 16+4g+(j-4)), followed by conv3's 40 channels from memory and the skip logits
// conv2 left in the S buffer.  No LDS.  Reads per full-resolution pixel: 16 (S) + 20 (conv3) + ~7 (quarter-res) bytes.
// ---------------------------------------------------------------------------------------------
struct Tail2 {
    const uint16_t* q0; const uint16_t* q1; int nq0, nq1;   // quarter-resolution sources of the inner deconv (chunks per pixel)
    const uint16_t* c3; int nc3;                            // half-resolution concat source of the outer deconv
    const float* S;                                         // skip logits [full-res canvas pixel][CP]
    int Hh, Wh, H0, W0, C;
    const uint16_t* wQ;            // inner deconv A fragments [ab][tile 2][k-step 4][64][8]
    const float* biasQ;            // [32]
    const uint16_t* wD;            // composed kernel, inner-deconv channels [tile][64][8]
    const uint16_t* wC;            // composed kernel, c3 channels [tile][k-step 2][64][8]
    const float* beta;
    float* out_logits; float* out_probs; int64_t* out_labels; uint8_t* out_labels_u8;
    float* out_margin;
};

constexpr int T2_ITER = 4;   // 16-pixel tiles a wave walks with one set of weight fragments in registers

template <int CP>
__global__ __launch_bounds__(256) void tail_fused2_kernel(Tail2 a) {
    constexpr int NTL = CP / 4;
    const int lane = threadIdx.x & 63, p16 = lane & 15, g = lane >> 4;
    // the four waves of a workgroup are the four sub-pixel parities (row hy & 1, column b) over the SAME quarter-resolution
    // pixels: the quarter-resolution sources are fetched from HBM once (the other three waves hit the CU's L1 / the XCD's L2)
    // instead of once per XCD that happens to own one of the parities
    const int w4 = threadIdx.x >> 6;
    const int tiles_x = (a.Wh + 31) >> 5, groups_x = (tiles_x + T2_ITER - 1) / T2_ITER;
    const int b = w4 & 1;
    const int qy = blockIdx.x / groups_x;
    const int hy = 2 * qy + (w4 >> 1);
    if (hy >= a.Hh) return;
    const int xt0 = (blockIdx.x - qy * groups_x) * T2_ITER;
    const int ab = (hy & 1) * 2 + b;
    // the weight fragments of this wave's sub-pixel stay in registers for all of its tiles (they were 11 of the 18
    // loads per tile: the kernel is bound by load issue, not by the 11 MFMAs)
    bf16x8 wq[4][2], wd[NTL], wc[NTL][2];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int t = 0; t < 2; ++t) wq[s][t] = *(const bf16x8*)(a.wQ + ((size_t)((ab * 2 + t) * 4 + s) * 64 + lane) * 8);
    float4 bq[2], bt[NTL];
#pragma unroll
    for (int t = 0; t < 2; ++t) bq[t] = *(const float4*)(a.biasQ + t * 16 + 4 * g);
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
        wd[t] = *(const bf16x8*)(a.wD + ((size_t)t * 64 + lane) * 8);
#pragma unroll
        for (int s = 0; s < 2; ++s) wc[t][s] = *(const bf16x8*)(a.wC + ((size_t)(t * 2 + s) * 64 + lane) * 8);
        bt[t] = *(const float4*)(a.beta + t * 16 + 4 * g);
    }
    typedef __attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int u32x4_t;
    constexpr unsigned OOBT = 0xfffffff0u;
    const bool two_q = a.nq1 > 0;
    const unsigned qpix = (unsigned)(a.Hh >> 1) * (unsigned)(a.Wh >> 1), hpix = (unsigned)a.Hh * (unsigned)a.Wh;
    const __amdgpu_buffer_rsrc_t rq0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.q0, 0, qpix * (unsigned)a.nq0 * 16u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rq1 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.q1 ? a.q1 : a.q0), 0, a.q1 ? qpix * (unsigned)a.nq1 * 16u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rc3 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.c3 ? a.c3 : a.q0), 0, a.c3 ? hpix * (unsigned)a.nc3 * 16u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc((void*)(a.S ? (const void*)a.S : (const void*)a.q0), 0, a.S ? hpix * (unsigned)(4 * CP * 4) : 0u, 0x00020000);
    const int C = a.C;
#pragma unroll 1
    for (int it = 0; it < T2_ITER; ++it) {
        const int x0 = (xt0 + it) * 32;
        if (x0 >= a.Wh) break;
        const int hxr = x0 + 2 * p16 + b;
        const bool inx = hxr < a.Wh;
        const int hx = inx ? hxr : a.Wh - 1;                 // clamp: loads stay in bounds, stores are masked
        const unsigned qpx = (unsigned)(hy >> 1) * (unsigned)(a.Wh >> 1) + (unsigned)(hx >> 1);
        const unsigned hpx = (unsigned)hy * (unsigned)a.Wh + (unsigned)hx;
        // ---- all loads first: buffer loads with 32-bit offsets; every load reads ONE tensor (two quarter-resolution sources: k-steps
        // 0-1 <- source 0, 2-3 <- source 1, the host packs the weights that way); a lane with nothing to fetch gets an out-of-range
        // offset and reads zeros -- no exec-mask branches, no 64-bit address arithmetic
        uint4 xq[4], xc[2];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            u32x4_t v;
            if (two_q) {
                const int ch = 4 * (s & 1) + g;
                if (s < 2) v = __builtin_amdgcn_raw_buffer_load_b128(rq0, ch < a.nq0 ? (qpx * (unsigned)a.nq0 + (unsigned)ch) * 16u : OOBT, 0, 0);
                else v = __builtin_amdgcn_raw_buffer_load_b128(rq1, ch < a.nq1 ? (qpx * (unsigned)a.nq1 + (unsigned)ch) * 16u : OOBT, 0, 0);
            } else {
                const int ch = 4 * s + g;
                v = __builtin_amdgcn_raw_buffer_load_b128(rq0, ch < a.nq0 ? (qpx * (unsigned)a.nq0 + (unsigned)ch) * 16u : OOBT, 0, 0);
            }
            xq[s] = make_uint4(v[0], v[1], v[2], v[3]);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int ch = 4 * s + g;
            const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rc3, ch < a.nc3 ? (hpx * (unsigned)a.nc3 + (unsigned)ch) * 16u : OOBT, 0, 0);
            xc[s] = make_uint4(v[0], v[1], v[2], v[3]);
        }
        f32x4 acc[NTL];
#pragma unroll
        for (int t = 0; t < NTL; ++t) {
            const int row0 = 16 * t + 4 * g, abo = row0 / CP, c0 = row0 - abo * CP;
            const unsigned px = (unsigned)(2 * hy + (abo >> 1)) * (unsigned)(2 * a.Wh) + (unsigned)(2 * hx + (abo & 1));
            const u32x4_t sv = __builtin_amdgcn_raw_buffer_load_b128(rS, (px * CP + c0) * 4u, 0, 0);   // fcn: no skip buffer (zero-length descriptor)
            const f32x4 svf = __builtin_bit_cast(f32x4, sv);   // (as ONE vector: element-wise bit casts of the loaded dwords made the compiler narrow the load to one dword)
            acc[t] = f32x4{bt[t].x + svf[0], bt[t].y + svf[1], bt[t].z + svf[2], bt[t].w + svf[3]};
        }
        // ---- inner deconv: d4[co][pixel] for this wave's sub-pixel ----
        // (bias = start value, as every MFMA conv of the engine: the stand-alone deconv kernel rounds the same way)
        f32x4 d4[2] = {f32x4{bq[0].x, bq[0].y, bq[0].z, bq[0].w}, f32x4{bq[1].x, bq[1].y, bq[1].z, bq[1].w}};
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int t = 0; t < 2; ++t)
                d4[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[s][t], __builtin_bit_cast(bf16x8, xq[s]), d4[t], 0, 0, 0);
        uint32_t pk[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            pk[t][0] = relu_pk_bf16(pk_bf16(d4[t][0], d4[t][1]), 0u);     // round, then ReLU on the packed pair (same value)
            pk[t][1] = relu_pk_bf16(pk_bf16(d4[t][2], d4[t][3]), 0u);
        }
        const uint4 xd = make_uint4(pk[0][0], pk[0][1], pk[1][0], pk[1][1]);
        // ---- composed deconv5 o logits ----
#pragma unroll
        for (int t = 0; t < NTL; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wd[t], __builtin_bit_cast(bf16x8, xd), acc[t], 0, 0, 0);
#pragma unroll
            for (int s = 0; s < 2; ++s)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[t][s], __builtin_bit_cast(bf16x8, xc[s]), acc[t], 0, 0, 0);
        }
        // ---- per pixel: classes live in CP/4 lane groups (as tail_composed_kernel) ----
#pragma unroll
        for (int t = 0; t < NTL; ++t) {
            const int row0 = 16 * t + 4 * g;
            const int abo = row0 / CP, c0 = row0 - abo * CP;
            const int y = 2 * hy + (abo >> 1), x = 2 * hx + (abo & 1);
            const bool inb = inx && y < a.H0 && x < a.W0;
            float bv, sv;
            int bi;
            top2_classes<CP / 4>(acc[t], c0, C, bv, bi, sv);
            const size_t p = (size_t)y * a.W0 + x;
            if (inb && c0 == 0) {
                if (a.out_labels_u8) a.out_labels_u8[p] = (uint8_t)bi;
                if (a.out_labels) a.out_labels[p] = bi;
                if (a.out_margin) a.out_margin[p] = bv - sv;
            }
            if (a.out_logits && inb)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (c0 + r < C) a.out_logits[p * C + c0 + r] = acc[t][r];
            if (a.out_probs) {
                float ex[4], sum = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) { ex[r] = (c0 + r < C) ? expf(acc[t][r] - bv) : 0.f; sum += ex[r]; }
                if constexpr (CP > 4) {
#pragma unroll
                    for (int sh = 16; sh < 4 * CP; sh <<= 1) sum += __shfl_xor(sum, sh);
                }
                if (inb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (c0 + r < C) a.out_probs[p * C + c0 + r] = ex[r] / sum;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Stand-alone logits layer (1x1 conv -> softmax / argmax) for graphs whose tail does not fuse
// (unet, res_unet): one MFMA GEMM per 16 pixels, rows = classes, K = channels; every B fragment is
// one coalesced 16-byte load per lane (no LDS).  HBM-bound: reads the last activation once.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void logits_mfma_kernel(const uint16_t* src0, int nch0, const uint16_t* src1, int nch1,
                                                          int Wp, int H0, int W0, int C, const uint16_t* wA /*[ks][64][8]*/,
                                                          const float* bias /*[16]*/, float* out_logits, float* out_probs,
                                                          int64_t* out_labels, uint8_t* out_labels_u8) {
    const int lane = threadIdx.x & 63, p16 = lane & 15, g = lane >> 4;
    const int wv = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int tiles_x = (W0 + 15) >> 4;
    const int y = wv / tiles_x, x = (wv - y * tiles_x) * 16 + p16;
    if (y >= H0) return;
    const bool inb = x < W0;
    const size_t px = (size_t)y * Wp + (inb ? x : 0);
    const int nks = (nch0 + nch1 + 3) >> 2;
    const float4 b = *(const float4*)(bias + 4 * g);
    f32x4 acc = f32x4{b.x, b.y, b.z, b.w};
    for (int s = 0; s < nks; ++s) {
        const int ch = 4 * s + g;
        uint4 xb = make_uint4(0, 0, 0, 0);
        if (ch < nch0) xb = *(const uint4*)(src0 + (px * nch0 + ch) * 8);
        else if (ch < nch0 + nch1) xb = *(const uint4*)(src1 + (px * nch1 + (ch - nch0)) * 8);
        const bf16x8 w = *(const bf16x8*)(wA + ((size_t)s * 64 + lane) * 8);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, __builtin_bit_cast(bf16x8, xb), acc, 0, 0, 0);
    }
    // lane (p16, g) holds classes 4g..4g+3 of pixel p16
    float bv = -3.4e38f;
    int bi = 0x7fffffff;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const bool take = (4 * g + r < C) & (acc[r] > bv);
        bv = take ? acc[r] : bv;
        bi = take ? 4 * g + r : bi;
    }
    if (C > 4) {
#pragma unroll
        for (int sh = 16; sh <= 32; sh <<= 1) {
            const float ov = __shfl_xor(bv, sh);
            const int oi = __shfl_xor(bi, sh);
            const bool take = (ov > bv) | ((ov == bv) & (oi < bi));
            bv = take ? ov : bv;
            bi = take ? oi : bi;
        }
    }
    const size_t p = (size_t)y * W0 + x;
    if (inb && g == 0) {
        if (out_labels_u8) out_labels_u8[p] = (uint8_t)bi;
        if (out_labels) out_labels[p] = bi;
    }
    if (out_logits && inb)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * g + r < C) out_logits[p * C + 4 * g + r] = acc[r];
    if (out_probs) {
        float ex[4], sum = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { ex[r] = (4 * g + r < C) ? expf(acc[r] - bv) : 0.f; sum += ex[r]; }
        if (C > 4) {
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
        }
        if (inb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * g + r < C) out_probs[p * C + 4 * g + r] = ex[r] / sum;
    }
}

// ---------------------------------------------------------------------------------------------
// first layer on MFMA (Cin = 1).  K = KS rows x 8 columns (columns >= KS carry zero weights):
// k-step s, lane group g <-> kernel row ky = 4s + g, element j <-> kernel column kx = j.  The B
// fragment of pixel (y, x) for row ky is therefore 8 consecutive input pixels
// t[y+ky][x .. x+7]: the uint8 tile is converted once (bf16(u * 1/255) == bf16(u / 255) for all
// 256 byte values, checked in tests/test_host_logic.py) and stored as 8 copies shifted by 0..7
// pixels, so that every fragment is one 16-byte-aligned ds_read_b128.
// ---------------------------------------------------------------------------------------------
template <int KS, int COUT>
__global__ __launch_bounds__(256) void conv1_mfma_kernel(const uint8_t* img, int H, int W, int Wp,
                                                         const uint16_t* wpk, const float* bias,
                                                         uint16_t* dst, int relu) {
    constexpr int CS = (COUT + 7) / 8 * 8, NT = (COUT + 15) / 16, NKS = (KS + 3) / 4, P = KS / 2;
    constexpr int TH1 = 16, TR = TH1 + KS - 1, TC = 48;      // tile rows, logical tile columns
    constexpr int RS = 80, CSTR = (TR * RS / 16 + ((2 - (TR * RS / 16) % 16) + 16) % 16) * 16;  // slots/copy = 2 (mod 16)
    __shared__ __attribute__((aligned(16))) uint16_t t[TR * TC];
    __shared__ __attribute__((aligned(16))) char cp[8 * CSTR];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p16 = lane & 15, g = lane >> 4;
    const int ox0 = blockIdx.x * 32, oy0 = blockIdx.y * TH1;
    // A fragments (weights) and biases of this lane
    bf16x8 wf[NKS][NT];
#pragma unroll
    for (int s = 0; s < NKS; ++s)
#pragma unroll
        for (int q = 0; q < NT; ++q) wf[s][q] = *(const bf16x8*)(wpk + ((size_t)(s * NT + q) * 64 + lane) * 8);
    float4 bv[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q) bv[q] = *(const float4*)(bias + q * 16 + 4 * g);
    // uint8 tile -> bf16(x/255), zero outside the image (pad-to-32 and SAME padding)
    for (int i = tid; i < TR * TC; i += 256) {
        const int r = i / TC, c = i - r * TC;
        const int y = oy0 + r - P, x = ox0 + c - P;
        float v = 0.0f;
        if (c < 32 + KS - 1 && y >= 0 && y < H && x >= 0 && x < W) v = (float)img[(size_t)y * W + x] * 0.00392156886f;
        t[i] = d_f2bf(v);
    }
    __syncthreads();
    // 8 shifted copies: copy c, row r, slot q holds t[r][8q + c .. 8q + c + 7]
    for (int i = tid; i < 8 * TR * 5; i += 256) {
        const int c = i / (TR * 5), rem = i - c * (TR * 5), r = rem / 5, q = rem - r * 5;
        const uint16_t* sp = t + r * TC + 8 * q + c;
        uint4 v;
        v.x = sp[0] | ((uint32_t)sp[1] << 16);
        v.y = sp[2] | ((uint32_t)sp[3] << 16);
        v.z = sp[4] | ((uint32_t)sp[5] << 16);
        v.w = sp[6] | ((uint32_t)sp[7] << 16);
        *(uint4*)(cp + c * CSTR + r * RS + q * 16) = v;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int row = wave * 4 + (m >> 1), xl = (m & 1) * 16 + p16;
        const char* base = cp + (xl & 7) * CSTR + (xl >> 3) * 16;
        f32x4 acc[NT];
#pragma unroll
        for (int q = 0; q < NT; ++q) acc[q] = f32x4{bv[q].x, bv[q].y, bv[q].z, bv[q].w};   // bias = start value (as the fused kernels)
#pragma unroll
        for (int s = 0; s < NKS; ++s) {
            const int ky = min(4 * s + g, KS - 1);   // rows past the kernel carry zero weights
            const bf16x8 xf = *(const bf16x8*)(base + (row + ky) * RS);
#pragma unroll
            for (int q = 0; q < NT; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s][q], xf, acc[q], 0, 0, 0);
        }
        const size_t o = ((size_t)(oy0 + row) * Wp + ox0 + xl) * CS;
#pragma unroll
        for (int q = 0; q < NT; ++q) {
            const int n = q * 16 + 4 * g;
            float v0 = acc[q][0], v1 = acc[q][1], v2 = acc[q][2], v3 = acc[q][3];
            if (relu) {
                v0 = v0 > 0.f ? v0 : 0.f; v1 = v1 > 0.f ? v1 : 0.f;
                v2 = v2 > 0.f ? v2 : 0.f; v3 = v3 > 0.f ? v3 : 0.f;
            }
            if (n < CS)
                *(uint2*)(dst + o + n) = make_uint2(pk_bf16(v0, v1),
                                                    pk_bf16(v2, v3));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// first layer, Cin = 1, wide output (unet 1 -> 64, res_unet 1 -> 32): the layer is a 128 (64) byte/px
// WRITE stream with 9 MACs per output, so the kernel is organised around full-line stores.  A wave
// owns 64 consecutive pixels of a row and walks down RB rows; lane = (pixel group pg, 8-channel
// chunk c): it computes PPL pixels x 8 couts and stores one 16-byte chunk per pixel, so the lanes of a
// pixel group write whole 128-byte lines.  The lane's KS*KS*8 weights stay in registers for the walk,
// the KS-row input window slides down one row per step (bytes -> bf16(x/255) on load).  Packed
// fp32 FMAs (two couts per instruction) in tap order over bf16-rounded operands: the same chain, hence
// the same bits, as conv1_bf16_kernel and the bf16-mode oracle.
// ---------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KS, int COUT, int RB>
__global__ __launch_bounds__(256) void conv1_rows_kernel(const uint8_t* __restrict__ img, int H, int W, int Hp, int Wp,
                                                         const float* __restrict__ w, const float* __restrict__ bias,
                                                         uint16_t* __restrict__ dst, int relu) {
    static_assert(COUT % 8 == 0 && (64 % (COUT / 8)) == 0, "couts in 8-channel chunks, a power of two of them");
    constexpr int NCH = COUT / 8, PGS = 64 / NCH, PPL = 64 / PGS, P = KS / 2, NX = PPL + KS - 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane % NCH, pg = lane / NCH;
    const int x0 = blockIdx.x * 64 + pg * PPL;                 // first pixel of this lane
    const int y0 = (blockIdx.y * 4 + wave) * RB;               // first row of this wave
    if (y0 >= Hp) return;
    f32x2 wv[KS * KS][4];
#pragma unroll
    for (int t = 0; t < KS * KS; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) wv[t][q] = *(const f32x2*)(w + t * COUT + 8 * c + 2 * q);
    f32x2 bv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) bv[q] = *(const f32x2*)(bias + 8 * c + 2 * q);
    bool colok[NX];
    unsigned coff[NX];
#pragma unroll
    for (int j = 0; j < NX; ++j) {
        const int x = x0 + j - P;
        colok[j] = x >= 0 && x < W;
        coff[j] = colok[j] ? (unsigned)x : 0u;
    }
    auto load_row = [&](int y, float* r) {
        const bool rowok = y >= 0 && y < H;                    // wave-uniform
        const uint8_t* rp = img + (size_t)(rowok ? y : 0) * (size_t)W;
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const float v = (float)rp[coff[j]] * 0.00392156886f;
            r[j] = (rowok && colok[j]) ? __uint_as_float((uint32_t)d_f2bf(v) << 16) : 0.0f;
        }
    };
    float win[KS][NX];
#pragma unroll
    for (int k = 0; k < KS - 1; ++k) load_row(y0 - P + k, win[k + 1]);      // rows y0-P .. y0+P-1 sit in slots 1..KS-1
    for (int r = 0; r < RB; ++r) {
        const int y = y0 + r;
        if (y >= Hp) break;
#pragma unroll
        for (int k = 0; k < KS - 1; ++k)
#pragma unroll
            for (int j = 0; j < NX; ++j) win[k][j] = win[k + 1][j];
        load_row(y + P, win[KS - 1]);
#pragma unroll
        for (int i = 0; i < PPL; ++i) {
            f32x2 acc[4] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
#pragma unroll
            for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
                    const float xv = win[ky][i + kx];
                    const f32x2 xx = f32x2{xv, xv};
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[q] = __builtin_elementwise_fma(xx, wv[ky * KS + kx][q], acc[q]);
                }
            uint32_t o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float a0 = acc[q][0] + bv[q][0], a1 = acc[q][1] + bv[q][1];
                if (relu) { a0 = vmax(a0, 0.f); a1 = vmax(a1, 0.f); }
                o[q] = pk_bf16(a0, a1);
            }
            const int x = x0 + i;
            if (x < Wp) *(uint4*)(dst + ((size_t)y * Wp + x) * COUT + 8 * c) = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
}

// generic input staging for graphs whose first conv runs on the MFMA kernel: uint8 -> bf16(x/255)
__global__ void preprocess_bf16_kernel(const uint8_t* img, int H, int W, int C, const float* lut,
                                       uint16_t* dst, int Hp, int Wp, int Cs) {
    const size_t n = (size_t)Hp * Wp;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(p % Wp), y = (int)(p / Wp);
        for (int c = 0; c < Cs; ++c) {
            float v = 0.0f;
            if (c < C && y < H && x < W) v = lut[img[((size_t)y * W + x) * C + c]];
            dst[p * Cs + c] = d_f2bf(v);
        }
    }
}

__global__ void pool_bf16_kernel(const uint16_t* in, int H, int W, int nch, uint16_t* out) {
    const size_t n = (size_t)(H / 2) * (W / 2) * nch;
    const int Wo = W / 2;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(t % nch);
        const size_t p = t / nch;
        const int x = (int)(p % Wo), y = (int)(p / Wo);
        const uint4* b = (const uint4*)(in + ((size_t)(2 * y) * W + 2 * x) * nch * 8) + ch;
        const uint4 m = max_bf16x8(max_bf16x8(b[0], b[nch]), max_bf16x8(b[(size_t)W * nch], b[(size_t)W * nch + nch]));
        ((uint4*)out)[t] = m;
    }
}

// ---------------------------------------------------------------------------------------------
// tail: crop + logits 1x1 + softmax + argmax, one pixel per thread (HBM-bound stream)
// ---------------------------------------------------------------------------------------------
template <int CMAX>
__global__ __launch_bounds__(256) void logits_bf16_kernel(const uint16_t* src0, int nch0, const uint16_t* src1,
                                                          int nch1, int Wp, int H, int W, const float* w,
                                                          const float* bias, int C, float* logits, float* probs,
                                                          int64_t* labels, uint8_t* labels_u8) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= H * W) return;
    const int y = p / W, x = p - y * W;
    const size_t q = (size_t)y * Wp + x;
    float z[CMAX];
#pragma unroll
    for (int c = 0; c < CMAX; ++c) z[c] = 0.0f;
    auto accum = [&](const uint16_t* src, int nch, int cbase) {
        const uint4* s = (const uint4*)(src + q * nch * 8);
        for (int ch = 0; ch < nch; ++ch) {
            const uint4 v = s[ch];
            const uint32_t ws[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xv = __uint_as_float((j & 1) ? (ws[j >> 1] & 0xffff0000u) : (ws[j >> 1] << 16));
                const float* wr = w + (size_t)(cbase + ch * 8 + j) * CMAX;
#pragma unroll
                for (int c = 0; c < CMAX; ++c) z[c] = __builtin_fmaf(xv, wr[c], z[c]);
            }
        }
    };
    accum(src0, nch0, 0);
    if (src1) accum(src1, nch1, nch0 * 8);
    int best = 0;
    float bv = 0.f;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
        if (c < C) {
            z[c] += bias[c];
            if (c == 0 || z[c] > bv) { bv = z[c]; best = c; }
        }
    }
    if (labels) labels[p] = best;
    if (labels_u8) labels_u8[p] = (uint8_t)best;
    if (logits)
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
            if (c < C) logits[(size_t)p * C + c] = z[c];
    if (probs) {
        float s = 0.0f;
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
            if (c < C) s += expf(z[c] - bv);
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
            if (c < C) probs[(size_t)p * C + c] = expf(z[c] - bv) / s;
    }
}

// ---------------------------------------------------------------------------------------------
// k5 stride-1 mid layers whose whole weight set fits LDS beside two input tiles (fcn / fcn_skip: conv3, conv4): a
// persistent PING-PONG kernel.  In conv_mfma_kernel a workgroup's life is prologue -> k-loop -> epilogue and only the
// k-loop feeds the matrix pipe; with three workgroups per CU the pipe is ~48 % busy (SQ_VALU_MFMA_BUSY_CYCLES), the
// rest of a wave's time goes to issuing weight / tile DMAs (the 75-96 KB weight set is re-streamed for every 8 x 32
// tile) and waiting for them.  Here ONE 768-thread workgroup per CU keeps the weights and the k-chunk table resident and
// walks its tiles in phases separated by one s_barrier:
//   * waves 8-11 (PRODUCERS) bring tile i + 1 into the other of two LDS tile buffers by LDS-DMA (zero fill outside the
//     image comes from the buffer descriptor) -- they never touch the matrix pipe;
//   * consumer team i & 1 (waves 0-3 or 4-7, one wave per SIMD each) runs the k-loop of tile i: the SIMD's matrix pipe
//     belongs to that wave for the phase;
//   * the other team stores the tile it accumulated in the previous phase (bias was the start value; fused 2x2 max-pool,
//     bf16 rounding) from its registers.
// A phase therefore lasts one k-loop; the epilogue and the DMA of the neighbouring tiles run beside it.
// The cout tile that is half empty (40 couts = 2.5 tiles) keeps only its 8 real rows in LDS (512-byte pieces; lanes of
// the empty rows read a zero slot): conv4's weights 96 -> 80 KB, which is what lets two 35 KB tiles fit.
// Same products, same k order, same start value as conv_mfma_kernel: the same bits.
// ---------------------------------------------------------------------------------------------
template <int SG, bool POOL>
__global__ __launch_bounds__(768) __attribute__((amdgpu_waves_per_eu(3, 3))) void conv_pp_kernel(MConv a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MT = 4, NT = 3, TH = 8, KS = 5, THH = TH + KS - 1, TWH = TW + KS - 1, PS2 = SG * 16;
    constexpr int WSTEP = 2 * 1024 + 512;                 // LDS bytes of one k-step's A fragments (third tile: rows 0-7 only)
    const int TB = a.lds_w_off;                           // bytes of one input tile (THH rows), 16-aligned (host)
    const int ks = a.ks_full;
    char* const w_t = smem + 2 * TB;
    char* const zero16 = w_t + ks * WSTEP;                // 16 zero bytes: the A rows that do not exist
    int* const tab_l = (int*)(zero16 + 16);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p16 = lane & 15, g = lane >> 4;
    const int tiles_x = (a.Wout + TW - 1) / TW;
    auto xcd_tile = [&](int t) {
        if (a.xq < 0) return t;
        const int x = t & 7, j = t >> 3;
        return x * a.xq + min(x, a.xr) + j;
    };
    auto origin = [&](int i, int& oy, int& ox) {
        const int t = xcd_tile((int)blockIdx.x + i * (int)gridDim.x);
        const int ty = t / tiles_x, tx = t - ty * tiles_x;
        oy = ty * TH; ox = tx * TW;
    };
    const int n_my = ((int)a.ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // tiles of this workgroup
    // ---- resident weights + k-chunk table + clean tile buffers: once per workgroup ---------------------------------
    for (int pc = wave; pc < ks * NT; pc += 12) {
        const int s_ = pc / NT, t_ = pc - s_ * NT;
        // packed A fragments: [k-step][cout tile][lane = (row p16, k-group g)][8]; tile 2 keeps lanes with p16 < 8 as slots g*8 + p16
        const int sl = t_ == 2 ? (lane >> 3) * 16 + (lane & 7) : lane;
        if (t_ < 2 || lane < 32)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.wpk + (size_t)pc * 512 + sl * 8),
                                             (__attribute__((address_space(3))) void*)(w_t + s_ * WSTEP + t_ * 1024), 16, 0, 0);
    }
    if (tid < ks * 4) tab_l[tid] = a.tab_full[tid];
    if (tid < 4) ((int*)zero16)[tid] = 0;
    for (int i = tid * 16; i < 2 * TB; i += 768 * 16) *(uint4*)(smem + i) = make_uint4(0, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();                                        // weights, table and the zeroed buffers are in place

    if (wave >= 8) {
        // =============================== PRODUCERS ===============================
        const int pw = wave - 8;
        const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.src0, 0, a.bytes0, 0x00020000);
        constexpr unsigned OOB = 0xfffffff0u;
        constexpr int ROW_SLOTS = TWH * SG, J = (ROW_SLOTS + 63) >> 6;
        const unsigned inv = 65536u / (unsigned)SG + 1u;
        auto stage = [&](int i) {
            int oy0, ox0;
            origin(i, oy0, ox0);
            const int iy0 = oy0 - a.pt, ix0 = ox0 - a.pl;
            char* const in_t = smem + (i & 1) * TB;
            unsigned col[J];
            bool live[J];
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int sl = j * 64 + lane;
                const int px = (int)(((unsigned)sl * inv) >> 16), cc = sl - px * SG;
                const int ix = ix0 + px;
                col[j] = (ix >= 0 && ix < a.Win && cc < a.nc_full) ? (unsigned)(ix * a.nch0 + cc) * 16u : OOB;   // (pad slot of a pixel: zeros)
                live[j] = sl < ROW_SLOTS;
            }
            for (int py = pw; py < THH; py += 4) {
                const int iy = iy0 + py;
                const bool rowv = iy >= 0 && iy < a.Hin;
                const unsigned rb = rowv ? (unsigned)iy * (unsigned)a.Win * (unsigned)(a.nch0 * 16) : OOB;
                char* drow = in_t + py * a.row_pitch;
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    const unsigned o = (col[j] == OOB || !rowv) ? OOB : rb + col[j];
                    if (live[j])
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (__attribute__((address_space(3))) void*)(drow + j * 1024), 16, o, 0, 0, 0);
                }
            }
        };
        if (n_my > 0) stage(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();                                    // tile 0 is in buffer 0
        for (int ph = 0; ph <= n_my; ++ph) {
            if (ph + 1 < n_my && !((a.dbg & 0x800) && ph >= 1)) stage(ph + 1);   // its buffer was read in phase ph - 1   (0x800: timing experiment, wrong results)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_barrier();
        }
        return;
    }
    // =============================== CONSUMERS ===============================
    const int team = wave >> 2, wv = wave & 3;            // the team's waves own two tile rows each
    float4 biasr[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) biasr[t] = *(const float4*)(a.bias + t * 16 + 4 * g);
    int pixbase[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) pixbase[m] = (wv * 2 + (m >> 1)) * a.row_pitch + ((m & 1) * 16 + p16) * PS2;
    const char* const wb0 = w_t + lane * 16;
    const char* const wb2 = p16 < 8 ? w_t + 2048 + (g * 8 + p16) * 16 : zero16 - 0;   // third cout tile: rows 8-15 do not exist
    const int w2step = p16 < 8 ? WSTEP : 0;
    const int* const tb = tab_l + g;
    const int CsO = a.nch_out * 8;
    constexpr unsigned OOBS = 0xfffffff0u;
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)a.dst, 0, a.dst_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)(POOL ? a.pool_dst : a.dst), 0, POOL ? a.pool_bytes : 0u, 0x00020000);
    f32x4 acc[MT][NT];
    int eoy = 0, eox = 0;                                 // origin of the tile held in acc
    lds_barrier();                                        // (pairs with the producers' "tile 0 is in buffer 0")
    for (int ph = 0; ph <= n_my; ++ph) {
        if ((ph & 1) == team) {
            if (ph < n_my) {
                // ---- k-loop of tile ph on buffer ph & 1 ----
                origin(ph, eoy, eox);
                const char* const in_t = smem + (ph & 1) * TB;
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{biasr[n].x, biasr[n].y, biasr[n].z, biasr[n].w};
                bf16x8 xa[MT], wa[NT], xb[MT], wbq[NT];
#define PP_LOAD(XF, WF, S, OFF)                                                                  \
                {                                                                                \
                    const int s_ = (S) < ks ? (S) : ks - 1;                                      \
                    WF[0] = *(const bf16x8*)(wb0 + s_ * WSTEP);                                  \
                    _Pragma("unroll") for (int m = 0; m < MT; ++m)                               \
                        XF[m] = *(const bf16x8*)(in_t + pixbase[m] + OFF);                       \
                    WF[1] = *(const bf16x8*)(wb0 + s_ * WSTEP + 1024);                           \
                    WF[2] = *(const bf16x8*)(wb2 + s_ * w2step);                                 \
                }
#define PP_TAB(S) tb[((S) < ks ? (S) : ks - 1) * 4]
#define PP_MMA(XF, WF)                                                                           \
                _Pragma("unroll") for (int t = 0; t < NT; ++t)                                   \
                    _Pragma("unroll") for (int m = 0; m < MT; ++m)                               \
                        acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WF[t], XF[m], acc[m][t], 0, 0, 0);
#define PP_INTERLEAVE                                                                            \
                _Pragma("unroll") for (int q_ = 0; q_ < MT * NT; ++q_) {                          \
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                            \
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                            \
                    __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);                            \
                }
                if (a.dbg & 0x400) __builtin_amdgcn_s_setprio(3);      // PSEG_PP_PRIO=1
                int offa = PP_TAB(0), offb = PP_TAB(1);
                PP_LOAD(xa, wa, 0, offa)
                int s = 0;
                for (; s + 2 <= ks; s += 2) {
                    __builtin_amdgcn_sched_barrier(0);
                    offa = PP_TAB(s + 2);
                    PP_LOAD(xb, wbq, s + 1, offb)
                    PP_MMA(xa, wa)
                    PP_INTERLEAVE
                    __builtin_amdgcn_sched_barrier(0);
                    offb = PP_TAB(s + 3);
                    PP_LOAD(xa, wa, s + 2, offa)
                    PP_MMA(xb, wbq)
                    PP_INTERLEAVE
                }
                __builtin_amdgcn_sched_barrier(0);
                if (ks & 1) { PP_MMA(xa, wa) }
                if (a.dbg & 0x400) __builtin_amdgcn_s_setprio(0);
#undef PP_INTERLEAVE
#undef PP_TAB
#undef PP_LOAD
#undef PP_MMA
            }
        } else if (ph >= 1 && !(a.dbg & 0x1000)) {
            // ---- epilogue of tile ph - 1 (accumulated by this team in the previous phase) ----   (0x1000: timing experiment)
            asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
            unsigned pixoff[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int y = eoy + wv * 2 + (m >> 1), x = eox + (m & 1) * 16 + p16;
                pixoff[m] = (y < a.Hout && x < a.Wout) ? (unsigned)(y * a.Wout + x) * (unsigned)(CsO * 2) : OOBS;
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int n = t * 16 + 4 * g;
                const unsigned noff = n < CsO ? (unsigned)n * 2u : OOBS;
                float v[MT][4];
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    v[m][0] = acc[m][t][0]; v[m][1] = acc[m][t][1]; v[m][2] = acc[m][t][2]; v[m][3] = acc[m][t][3];
                    const unsigned o = (pixoff[m] == OOBS || noff == OOBS) ? OOBS : pixoff[m] + noff;
                    if (a.relu) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[m][r] = vmax(v[m][r], 0.0f);
                    }
                    const uint2 pk = make_uint2(pk_bf16(v[m][0], v[m][1]), pk_bf16(v[m][2], v[m][3]));
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, pk), rd, o, 0, 0);
                }
                if (POOL) {
                    const int Wo2 = a.Wout >> 1, Ho2 = a.Hout >> 1;
#pragma unroll
                    for (int m = 0; m < 2; ++m) {             // pairs (m, m + 2): rows 2r and 2r + 1 of this wave
                        float q[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) q[r] = vmax_xor1(vmax(v[m][r], v[m + 2][r]));
                        const int y = (eoy >> 1) + wv;
                        const int x = (eox >> 1) + ((m * 16 + p16) >> 1);
                        const bool ok = !(p16 & 1) && y < Ho2 && x < Wo2 && noff != OOBS;
                        const unsigned o = ok ? (unsigned)(y * Wo2 + x) * (unsigned)(CsO * 2) + noff : OOBS;
                        const uint2 pk = make_uint2(pk_bf16(q[0], q[1]), pk_bf16(q[2], q[3]));
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, pk), rp, o, 0, 0);
                    }
                }
            }
        }
        lds_barrier();
    }
}

// ---------------------------------------------------------------------------------------------
// k5 stride-1 layers whose weight set does NOT fit LDS (fcn / fcn_skip: conv5, conv6, conv7, deconv1 (+ deconv2), deconv3):
// STREAMED weights, PERSISTENT workgroups, dedicated LOADER waves -- conv_sp_kernel.
// In conv_mfma_kernel every wave of a workgroup does everything in turn: issue the tile DMA, issue a weight group, wait,
// s_barrier, a few k-steps, again -- s_memtime stamps of the 1/8-resolution layers (2048x1536, one 8 x 32 tile per CU): 54 k
// cycles per workgroup for 16 k cycles of matrix work; 7 k of them ISSUING weight DMAs from the compute waves, 5 k in group
// waits, a drained software pipeline at each of the 14 group barriers, the tile re-staged per channel block, a serial
// prologue (8.5 k) and epilogue (6 k; 20 k where the k2 s2 transposed conv follows on the accumulators, its A fragments
// fetched from L2 one group ahead).  The fill rate is NOT the limit: four waves with four requests in flight each stream
// the shared 256 KB weight set at 55 B/clk/CU (tools/microtests/lds_dma_ring.hip; the chip's L2 gives 24 TB/s).
// Here ONE 512-thread workgroup per CU owns all 160 KB of LDS and splits the roles:
//   * waves 0-3 (one per SIMD) COMPUTE: one flat, continuously software-pipelined k-loop over all channel blocks of a tile
//     (no barrier, no wait, no DMA instruction inside), then the epilogue, then the next tile of the workgroup;
//   * waves 4-5 stream the packed weights through a ring of RK k-steps (NT KiB each), as far ahead as the ring allows;
//   * waves 6-7 stage the halo tiles, one channel block per LDS slot (all blocks of a tile resident at once; the next
//     tile's block b lands while the current tile is past its own block b).
// There is no s_barrier after set-up.  Loaders and compute waves meet through monotone counters in LDS: `ready` / `tready`
// (weight groups / tile blocks landed: written by a loader after its counted s_waitcnt vmcnt), `done` / `tdone` (groups /
// blocks a compute wave has finished reading).  A compute wave reads the loaders' counters at the START of a k-step's
// region and looks at the value at its END (the LDS latency hides under the MFMAs); only a late loader sends it into a
// polling loop.  Every polling loop is bounded (SP_SPIN_LIMIT polls, then it gives up, records the fact in SConv::err and
// goes on with whatever is in LDS): a protocol bug produces wrong numbers, not a hung GPU.
// Same products in the same k order with the same start value (bias) as conv_mfma_kernel on the dense tile: the same bits.
// The Conv2DTranspose k2 s2 behind deconv1 (FL_DQ) takes its 48 A fragments from the SAME ring (they follow the conv's
// k-steps in the packed stream), so its epilogue never waits for L2.
// A tile index carries the page: tile t -> page t / tiles_per_page -- a launch covers all pages of a batch (pseg_predict_batch).
// ---------------------------------------------------------------------------------------------
struct SConv {
    const uint16_t* src0; const uint16_t* src1;
    int nch0, nch1;                  // 16-byte chunks per pixel of the two (concatenated) sources
    unsigned bytes0, bytes1;         // bytes of ONE page of each source
    int Hin, Win, Hout, Wout, pt, pl, relu;
    int nblk, nc_full, nc_last;      // channel blocks of the tile: nblk slots of TBLK bytes, nc chunks each
    int K;                           // conv k-steps per tile (all blocks; even)
    int S;                           // steps of the weight stream per tile: K (+ the transposed conv's pseudo-steps), a multiple of SP_GK
    int blk_steps;                   // k-steps per full channel block
    const int* tab;                  // [K * 4] byte offsets of the k-chunks inside the tile region (block slot included)
    const uint16_t* wpk;             // [S][NT][64][8] bf16
    const float* bias;               // [NT * 16]
    int row_pitch, TBLK, RK;         // LDS: tile row pitch, bytes of a block slot, ring capacity in steps
    int lds_ring_off, lds_patch_off, lds_tab_off, lds_flag_off;
    uint16_t* dst; unsigned dst_bytes; int nch_out;          // dst_bytes: ONE page (0: the tensor is not stored)
    uint16_t* pool_dst; unsigned pool_bytes;
    const float* dq_bias; uint16_t* dq_dst; unsigned dq_bytes; int dq_nch, dq_relu;
    int ntiles, tiles_per_page, xq, xr;
    int* err;                        // != 0 after a polling loop gave up: the engine's 8-int record (Engine::d_sp_err), read -- and cleared -- by engine_status()
    int layer_id;                    // index of the op in Engine::ops, for that record
    unsigned long long* trace;       // PSEG_SP_TRACE=<layer>: 16 s_memtime stamps per workgroup, or null
    int dbg;                         // diagnostic build only (PSEG_SP_DBG, wrong results): 1 no weight DMA, 2 no tile DMA (timing only); 8 the weight loaders stop after their
                                     // first groups and every wait gives up after 64 polls -- the give-up path itself, for tests/test_bf16_gpu.py
};
enum { SP_POOL = 1, SP_DQ = 2, SP_PATCH = 4 };
constexpr int SP_GK = 2;             // k-steps per weight group (the unit of the ready / done counters)
// weight loaders publish group q once group q + lag has been issued (counted vmcnt): lag = min(3, ring groups - 3) -- a compute
// wave inside group c must already see group c + 1 (loader at q >= c + lag + 1) while the ring lets the loader reach q <= c + NS - 1
constexpr int SP_SPIN_LIMIT = 1 << 14;   // polls (~100-200 cycles each) before a waiter gives up: > 1 ms, a real wait is microseconds
constexpr int SP_DQ_PIECES = 48;     // A fragments of the fused transposed conv: [ab 4][cout tile 4][k-step 3]

// tile row pitch of conv_sp_kernel for a pixel of SG 16-byte slots and a tile of `tw` output pixels per row: tw + 4 pixels + the
// pad the bank model picks for the 32-wide tile (pair_chunks; the host checks that it still does) -- a compile-time constant so
// that fragment addresses are one register + immediates
__host__ __device__ constexpr int sp_row_pitch(int sg, int tw = 32) { return sg == 4 ? (tw + 4) * 64 : (sg == 5 ? (tw + 4) * 80 + 48 : 0); }

template <int N>
__device__ __forceinline__ void sp_wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// TWK = output pixels per tile row: 32 (a wave's two rows are four 16-pixel tiles) or 24 (three: two full ones and one whose lanes
// 0-7 / 8-15 are columns 16-23 of the first / second row).  The narrow tile exists for launches of ONE round: a 2048x1536 page has
// 256 x 192 pixels at 1/8 resolution = 192 tiles of 8 x 32 for 256 CUs, but exactly 256 of 8 x 24 -- every CU gets three quarters of
// the work instead of a quarter of the chip idling (host: sp_pick_tw).
template <int NT, int SG, int FL, int TWK = 32>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_sp_kernel(SConv a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(TWK == 32 || TWK == 24, "tile rows of two or one and a half 16-pixel MFMA tiles");
    constexpr int TW = TWK;                                   // (shadows the engine-wide 32)
    constexpr int MT = TWK == 32 ? 4 : 3, TH = 8, KS = 5, THH = TH + KS - 1, TWH = TW + KS - 1, PS2 = SG * 16;
    constexpr int ROWP = sp_row_pitch(SG, TWK);               // tile row pitch: a constant, so that a wave's pixel fragments are one address + immediates
    constexpr int WSTEP = NT * 1024;
    constexpr bool POOL = (FL & SP_POOL) != 0, DQ = (FL & SP_DQ) != 0, PATCH = (FL & SP_PATCH) != 0;
    static_assert(!DQ || NT == 5, "the fused transposed conv is written for 80 channels");
    static_assert(!(POOL && TWK != 32), "the fused pool pairs the rows of full 16-pixel tiles");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* const ring = smem + a.lds_ring_off;
    // the counters: [0,1] ready  [2,3] tready  [4..7] done  [8..11] tdone.  Plain LDS accesses through address-space-3 pointers (a
    // `volatile` access became a FLAT load with a full vmcnt(0) wait behind it -- in the loaders that drained their own DMA queue
    // at every poll); re-reads are forced by the compiler barriers (asm memory clobbers) around every use instead.
    typedef __attribute__((address_space(3))) int lds_int;
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    lds_int* const flags = (lds_int*)(smem + a.lds_flag_off);
    const __attribute__((address_space(3))) i32x4* const flagsv = (const __attribute__((address_space(3))) i32x4*)(smem + a.lds_flag_off);   // the loaders' four counters in one read
    const int tiles_x = (a.Wout + TW - 1) / TW;
    const int n_my = ((int)a.ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // tiles of this workgroup (>= 1)
    const bool one_blk = a.nblk == 1;               // single channel block: two slots, tiles alternate
    const int nslot = one_blk ? 2 : a.nblk;
    auto xcd_tile = [&](int t) {
        if (a.xq < 0) return t;
        const int x = t & 7, j = t >> 3;
        return x * a.xq + min(x, a.xr) + j;
    };
    auto origin = [&](int i, int& oy, int& ox, int& pg) {
        const int t = xcd_tile((int)blockIdx.x + i * (int)gridDim.x);
        pg = t / a.tiles_per_page;
        const int tl = t - pg * a.tiles_per_page;
        const int ty = tl / tiles_x, tx = tl - ty * tiles_x;
        oy = ty * TH; ox = tx * TW;
    };
    // PSEG_SP_TRACE (developer aid): s_memtime stamps of wave 0 / 4 / 6, 16 slots per workgroup (tools/sp_trace.py)
    unsigned long long* const trc = (PSEG_DIAG && a.trace) ? a.trace + (size_t)blockIdx.x * 16 : nullptr;   // (diagnostic build only: the release kernel holds no stamp)
#define SP_STAMP(i) if (trc && lane == 0) trc[i] = __builtin_amdgcn_s_memtime();
    if (trc && lane == 0) { unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); ((unsigned short*)&trc[14])[wave] = (unsigned short)hw; }   // where the wave runs: bits 5:4 = SIMD
    // the FIRST wait that gave up leaves its code and what it was waiting for: err[0] code, [1] wave, [2] needed, [3] had, [4] second need, [5] second had
    auto give_up = [&](int code, int need, int have, int need2 = 0, int have2 = 0) {
        if (lane == 0 && atomicCAS(a.err, 0, code) == 0) { a.err[1] = wave; a.err[2] = need; a.err[3] = have; a.err[4] = need2; a.err[5] = have2; a.err[6] = (int)blockIdx.x; a.err[7] = a.layer_id; }
    };
    const int spin_limit = (PSEG_DIAG && (a.dbg & 8)) ? 64 : SP_SPIN_LIMIT;
    // every wave clears the counters it writes, then the only barrier of the kernel: nothing else is shared before it
    if (lane == 0) {
        if (wave < 4) { flags[4 + wave] = 0; flags[8 + wave] = 0; }
        else flags[wave - 4] = 0;
    }
    if (wave == 0) { SP_STAMP(0) }
    lds_barrier();

    if (wave >= 6) {
        // =============================== TILE LOADERS ===============================
        const int tw = wave - 6;
        constexpr unsigned OOB = 0xfffffff0u;
        constexpr int ROW_SLOTS = TWH * SG, J = (ROW_SLOTS + 63) >> 6;
        constexpr int PER_BLOCK = (THH / 2) * J;              // DMA instructions of one wave per channel block (a block has ONE source: host)
        static_assert(THH % 2 == 0 && 3 * PER_BLOCK < 64, "counted vmcnt waits of the tile loaders");
        const unsigned inv = 65536u / (unsigned)SG + 1u;
        const int nblocks = n_my * a.nblk;
        if (tw == 0) {
            // the k-chunk table first (K * 16 bytes in 1 KiB pieces): whoever sees the first tile block landed also sees the table
            const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc((void*)a.tab, 0, (unsigned)a.K * 16u, 0x00020000);
            for (int pc = 0; pc * 1024 < a.K * 16; ++pc)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rt, (__attribute__((address_space(3))) void*)(smem + a.lds_tab_off + pc * 1024), 16, (unsigned)(pc * 1024 + lane * 16), 0, 0, 0);
        }
        auto stage = [&](int jb) {
            if (PSEG_DIAG && (a.dbg & 2)) return;
            const int i = jb / a.nblk, b = jb - i * a.nblk;
            int oy0, ox0, pg;
            origin(i, oy0, ox0, pg);
            const int iy0 = oy0 - a.pt, ix0 = ox0 - a.pl;
            char* const in_t = smem + (one_blk ? (i & 1) : b) * a.TBLK;
            const int c0 = b * a.nc_full, nc = b == a.nblk - 1 ? a.nc_last : a.nc_full;
            const bool s1 = c0 >= a.nch0;                     // (wave-uniform) the block's chunks come from the second source
            const int nchs = s1 ? a.nch1 : a.nch0, cs0 = s1 ? c0 - a.nch0 : c0;
            const unsigned pbytes = s1 ? a.bytes1 : a.bytes0;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)(s1 ? a.src1 : a.src0) + (size_t)pg * pbytes), 0, pbytes, 0x00020000);
            unsigned col[J];
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int sl = j * 64 + lane;
                const int px = (int)(((unsigned)sl * inv) >> 16), cc = sl - px * SG;
                const int ix = ix0 + px;
                col[j] = (cc < nc && ix >= 0 && ix < a.Win && sl < ROW_SLOTS) ? (unsigned)(ix * nchs + cs0 + cc) * 16u : OOB;
            }
            const unsigned rowb = (unsigned)a.Win * (unsigned)(nchs * 16);
            for (int py = tw; py < THH; py += 2) {
                const int iy = iy0 + py;
                const bool rowv = iy >= 0 && iy < a.Hin;
                const unsigned rb = rowv ? (unsigned)iy * rowb : OOB;
                char* drow = in_t + py * ROWP;
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    // (slots past the row end belong to the next row's start: a zero lands there before that row's own load when
                    // the rows go in order; this wave takes every other row, so those lanes are switched off instead)
                    const unsigned o = (col[j] == OOB || !rowv) ? OOB : rb + col[j];
                    if (j * 64 + lane < ROW_SLOTS)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(drow + j * 1024), 16, o, 0, 0, 0);
                }
            }
        };
        // Every workgroup of the launch starts at the same moment and the chip delivers ~11 B/clk/CU to such a burst: what the
        // first k-steps need goes first and alone -- the table, block 0 (and, in the weight loaders, two groups) -- the other
        // blocks of the first tile follow once it has landed (they are needed a channel block = ~25 k-steps later).
        const int pre = min(nslot, nblocks);
        stage(0);
        sp_wait_vmcnt<0>();
        if (lane == 0) flags[2 + tw] = 1;
        for (int jb = 1; jb < pre; ++jb) stage(jb);           // the slots are empty: no counter to look at
        // publish them as they land (in order; a block = PER_BLOCK instructions of this wave)
        if (pre > 3) { sp_wait_vmcnt<2 * PER_BLOCK>(); if (lane == 0) flags[2 + tw] = pre - 2; }
        if (pre > 2) { sp_wait_vmcnt<PER_BLOCK>(); if (lane == 0) flags[2 + tw] = pre - 1; }
        sp_wait_vmcnt<0>();
        if (lane == 0) flags[2 + tw] = pre;
        for (int jb = pre; jb < nblocks; ++jb) {
            // the slot is free once every compute wave is past block jb - nslot
            const int need = jb - nslot + 1;
            for (int it = 0;; ++it) {
                asm volatile("" ::: "memory");
                const int d0 = flags[8], d1 = flags[9], d2 = flags[10], d3 = flags[11];
                if (__builtin_amdgcn_readfirstlane(min(min(d0, d1), min(d2, d3))) >= need) break;
                if (it >= spin_limit) { give_up(1, need, min(min(d0, d1), min(d2, d3)), jb); break; }
                __builtin_amdgcn_s_sleep(2);
            }
            asm volatile("" ::: "memory");
            stage(jb);
            sp_wait_vmcnt<0>();
            if (lane == 0) flags[2 + tw] = jb + 1;
        }
        if (tw == 0) { SP_STAMP(13) }
        return;
    }
    if (wave >= 4) {
        // =============================== WEIGHT LOADERS ===============================
        const int lw = wave - 4;
        const int gpt = a.S / SP_GK;                          // groups per tile
        const int total = n_my * gpt;
        const int NS = a.RK / SP_GK;                          // ring slots (groups), >= 5 (host)
        const int lag = min(3, NS - 4);
        // group q: 2 NT pieces of 1 KiB; this wave takes pieces lw, lw + 2, ... (NT of them)
        // (buffer form: the per-lane offset is ONE constant register, the piece's offset rides in the scalar operand -- no
        // per-load address arithmetic, half the address registers of the global form)
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpk, 0, (unsigned)a.S * (unsigned)(NT * 1024), 0x00020000);
        const unsigned vo = (unsigned)lane * 16u;
        int qt = 0, slot = 0;                                 // group within the tile's stream, ring slot (both of the NEXT group to issue)
        auto issue = [&]() {
            const unsigned so = (unsigned)(qt * (SP_GK * NT) + lw) * 1024u;
            char* dstb = ring + (slot * (SP_GK * NT) + lw) * 1024;
            if (!(PSEG_DIAG && (a.dbg & 1))) {
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(dstb + j * 2048), 16, vo, so + (unsigned)j * 2048u, 0, 0);
            }
            qt = qt + 1 == gpt ? 0 : qt + 1;
            slot = slot + 1 == NS ? 0 : slot + 1;
        };
        const int pre = min(NS, total);
        const int pre0 = min(2, pre);
        for (int q = 0; q < pre0; ++q) issue();               // the ring is empty: two groups at once ...
        sp_wait_vmcnt<0>();
        if (lane == 0) flags[lw] = pre0;
        for (int it = 0; it < spin_limit; ++it) {             // ... the rest behind the first tile block (see the tile loaders)
            asm volatile("" ::: "memory");
            const int t0 = flags[2], t1 = flags[3];
            if (__builtin_amdgcn_readfirstlane(min(t0, t1)) >= 1) break;
            __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
        for (int q = pre0; q < pre; ++q) issue();
        // publish what is in flight as it lands: at most 6 more groups were issued (NS <= 8: host)
#define SP_PUB(REM)                                                                              \
        if (pre - pre0 > (REM)) { sp_wait_vmcnt<(REM) * NT>(); if (lane == 0) flags[lw] = pre - (REM); }
        SP_PUB(5) SP_PUB(4) SP_PUB(3) SP_PUB(2) SP_PUB(1) SP_PUB(0)
#undef SP_PUB
        int pub = pre;                                        // groups published so far (everything issued above has landed)
        long long wl_poll = 0;
        if (PSEG_DIAG && (a.dbg & 8)) return;                 // (diagnostic build: a weight loader that stops -- the compute waves' waits give up)
        for (int q = pre; q < total; ++q) {
            const int need = q - NS + 1;                      // every compute wave has finished group q - NS
            const long long tp0 = trc ? __builtin_amdgcn_s_memtime() : 0;
            for (int it = 0;; ++it) {
                asm volatile("" ::: "memory");
                const int d0 = flags[4], d1 = flags[5], d2 = flags[6], d3 = flags[7];
                if (__builtin_amdgcn_readfirstlane(min(min(d0, d1), min(d2, d3))) >= need) break;
                // the ring is full: what has been issued lands while this wave waits -- publish it now, not `lag` groups behind the next issue
                if (pub < q) { sp_wait_vmcnt<0>(); pub = q; if (lane == 0) flags[lw] = pub; }
                if (it >= spin_limit) { give_up(2, need, min(min(d0, d1), min(d2, d3)), q); break; }
                __builtin_amdgcn_s_sleep(1);
            }
            asm volatile("" ::: "memory");
            if (trc) wl_poll += __builtin_amdgcn_s_memtime() - tp0;
            issue();
            // groups <= q - lag have landed once at most `lag` groups of this wave's loads are outstanding
            if (lag == 3) sp_wait_vmcnt<3 * NT>(); else if (lag == 2) sp_wait_vmcnt<2 * NT>(); else sp_wait_vmcnt<NT>();
            if (q - lag + 1 > pub) { pub = q - lag + 1; if (lane == 0) flags[lw] = pub; }
        }
        sp_wait_vmcnt<0>();
        if (lane == 0) flags[lw] = total;
        if (lw == 0) { SP_STAMP(11) if (trc && lane == 0) trc[12] = (unsigned long long)wl_poll; }
        return;
    }
    // =============================== COMPUTE ===============================
    if (wave == 0) { SP_STAMP(1) }
    long long sw_cyc = 0; int sw_cnt = 0;
    const int p16 = lane & 15, g = lane >> 4;
    float4 biasr[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) biasr[t] = *(const float4*)(a.bias + t * 16 + 4 * g);
    // pixel tile m of this wave: row wave * 2 + PR(m, p), column PC(m, p) for lane pixel p (TWK 32: uniform per tile -- immediates)
    auto PR = [](int m, int p) { return TWK == 32 ? (m >> 1) : (m < 2 ? m : (p >> 3)); };
    auto PC = [](int m, int p) { return TWK == 32 ? (m & 1) * 16 + p : (m < 2 ? p : 16 + (p & 7)); };
    const int pixb0 = (wave * 2) * ROWP + p16 * PS2;          // TWK 32, fragment m: + (m >> 1) * ROWP + (m & 1) * 16 * PS2 (immediates)
    int pixm[MT];                                             // TWK 24: the lane's own offset per tile (the third tile straddles the two rows)
#pragma unroll
    for (int m = 0; m < MT; ++m) pixm[m] = (wave * 2 + PR(m, p16)) * ROWP + PC(m, p16) * PS2;
    const char* const wb0 = ring + lane * 16;
    const char* const tb = smem + a.lds_tab_off + g * 4;
    const int K = a.K, RK = a.RK;
    const int CsO = a.nch_out * 8;
    constexpr unsigned OOBS = 0xfffffff0u;

    // slow path of the counters: poll until `needw` weight groups and `needt` tile blocks have landed
    auto slow_wait = [&](int needw, int needt) {
        const long long tq0 = trc ? __builtin_amdgcn_s_memtime() : 0;
        for (int it = 0;; ++it) {
            asm volatile("" ::: "memory");
            const int r0 = flags[0], r1 = flags[1], t0 = flags[2], t1 = flags[3];
            if (__builtin_amdgcn_readfirstlane(min(r0, r1)) >= needw && __builtin_amdgcn_readfirstlane(min(t0, t1)) >= needt) break;
            if (it >= spin_limit) { give_up(3, needw, min(r0, r1), needt, min(t0, t1)); break; }
            __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
        if (trc) { sw_cyc += __builtin_amdgcn_s_memtime() - tq0; ++sw_cnt; }
    };
    int kg = 0;                       // weight-stream steps consumed by the tiles before this one (even)
    int pos0 = 0;                     // ring position of this tile's step 0
    for (int i = 0; i < n_my; ++i) {
        int oy0, ox0, pg;
        origin(i, oy0, ox0, pg);
        const int jb0 = i * a.nblk;                           // tile blocks before this tile
        const char* const in_t = smem + (TWK == 32 ? pixb0 : 0) + (one_blk ? (i & 1) * a.TBLK : 0);
        f32x4 acc[MT][NT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{biasr[n].x, biasr[n].y, biasr[n].z, biasr[n].w};
        // ---- the flat k-loop: one trip = one weight group = two k-steps ----------------------------------------------
        // Needs of step x of this tile: weight group (kg + x) / 2 landed (ready >= that + 1), tile block x / blk_steps landed
        // (tready >= jb0 + that + 1).  Entering a trip the needs of its steps s, s + 1 and of step s + 2 (whose fragments the
        // trip's second half requests) have been looked at; behind its MFMAs the trip looks at those of s + 3 and s + 4.
        bf16x8 xa[MT], wa[NT], xb[MT], wbq[NT];
#define SP_LOAD(XF, WF, POS, OFF)                                                                 \
        {                                                                                        \
            const char* wbp_ = wb0 + (POS) * WSTEP;                                              \
            const char* xp_ = in_t + (OFF);                                                      \
            WF[0] = *(const bf16x8*)(wbp_);                                                      \
            _Pragma("unroll") for (int m = 0; m < MT; ++m)                                       \
                XF[m] = *(const bf16x8*)(xp_ + (TWK == 32 ? (m >> 1) * ROWP + (m & 1) * 16 * PS2 : pixm[m])); \
            _Pragma("unroll") for (int t = 1; t < NT; ++t)                                       \
                WF[t] = *(const bf16x8*)(wbp_ + t * 1024);                                       \
        }
#define SP_MMA(XF, WF)                                                                           \
        _Pragma("unroll") for (int t = 0; t < NT; ++t)                                           \
            _Pragma("unroll") for (int m = 0; m < MT; ++m)                                       \
                acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WF[t], XF[m], acc[m][t], 0, 0, 0);
#define SP_INTERLEAVE                                                                            \
        _Pragma("unroll") for (int q_ = 0; q_ < MT * NT; ++q_) {                                  \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                    \
            __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);                                    \
        }
#define SP_TAB(X) (*(const int*)(tb + ((X) < K ? (X) : K - 1) * 16))
        static_assert(SP_GK == 2, "a weight group is the two k-steps of one trip of the k-loop");
        // tile blocks: the block of the last step whose needs have been looked at, and the blocks the k-loop is past
        int blk4 = 0, blk4_next = a.nblk > 1 ? a.blk_steps : 0x7fffffff;   // first step of block blk4 + 1
        int rel_blk = 0, rel_next = a.nblk > 1 ? a.blk_steps : 0x7fffffff;  // first step of block rel_blk + 1
        {
            const int x2 = min(2, K - 1);                     // steps 0, 1, 2: requested before the first check
            while (x2 >= blk4_next) { ++blk4; blk4_next = blk4 + 1 < a.nblk ? blk4_next + a.blk_steps : 0x7fffffff; }
            slow_wait(((kg + x2) >> 1) + 1, jb0 + blk4 + 1);
        }
        if (wave == 0 && i < 2) { SP_STAMP(2 + 3 * i) }
        int pos1 = pos0 + 1 == RK ? 0 : pos0 + 1, pos2 = pos1 + 1 == RK ? 0 : pos1 + 1;   // ring positions of steps s + 1, s + 2
        int offa = SP_TAB(0), offb = SP_TAB(1);
        SP_LOAD(xa, wa, pos0, offa)
        for (int s = 0; s < K; s += 2) {                      // K is even
            __builtin_amdgcn_sched_barrier(0);
            const i32x4 fv = *flagsv;                         // looked at behind the trip's MFMAs
            offa = SP_TAB(s + 2);
            SP_LOAD(xb, wbq, pos1, offb)
            SP_MMA(xa, wa)
            SP_INTERLEAVE
            __builtin_amdgcn_sched_barrier(0);
            offb = SP_TAB(s + 3);
            SP_LOAD(xa, wa, pos2, offa)
            SP_MMA(xb, wbq)
            SP_INTERLEAVE
            __builtin_amdgcn_sched_barrier(0);
            // the group is read (its fragments are in registers: the MFMAs were issued): hand it back, and the tile blocks behind
            if (lane == 0) flags[4 + wave] = ((kg + s) >> 1) + 1;
            if (s + 2 >= rel_next) {
                ++rel_blk; rel_next = rel_blk + 1 < a.nblk ? rel_next + a.blk_steps : 0x7fffffff;
                if (lane == 0) flags[8 + wave] = jb0 + rel_blk;
            }
            // needs of steps s + 3 and s + 4 (requested by the next trip)
            const int x4 = min(s + 4, K - 1);
            if (x4 >= blk4_next) { ++blk4; blk4_next = blk4 + 1 < a.nblk ? blk4_next + a.blk_steps : 0x7fffffff; }
            const int needw = ((kg + x4) >> 1) + 1, needt = jb0 + blk4 + 1;
            if (__builtin_amdgcn_readfirstlane(min(fv.x, fv.y)) < needw || __builtin_amdgcn_readfirstlane(min(fv.z, fv.w)) < needt)
                slow_wait(needw, needt);
            asm volatile("" ::: "memory");
            pos1 = pos2 + 1 == RK ? 0 : pos2 + 1;
            pos2 = pos1 + 1 == RK ? 0 : pos1 + 1;
        }
        __builtin_amdgcn_sched_barrier(0);
#undef SP_TAB
#undef SP_INTERLEAVE
#undef SP_MMA
#undef SP_LOAD
        if (lane == 0) flags[8 + wave] = jb0 + a.nblk;        // the tile's last block: read to the end
        if (wave == 0 && i < 2) { SP_STAMP(3 + 3 * i) }
        // ---- epilogue from registers ---------------------------------------------------------------------------------
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
        if constexpr (DQ) {
            // Conv2DTranspose k2 s2 (ReLU) on the accumulators (see conv_mfma_kernel FL_DQ): a 16 x 16 accumulator tile has the pixel
            // on the lane and four channels in registers, so two tiles are one B operand of the next MFMA.  Its 48 A fragments are
            // pieces K * NT ... of this tile's weight stream (ring steps pos0 + K ...; piece p at step p / NT, KiB p % NT), so
            // nothing is fetched from L2 here.  Per sub-pixel ab and pixel tile m the 64 output channels of 16 pixels (12 MFMAs) go
            // through a 2 KiB LDS patch of this wave and leave as whole 128-byte lines -- a direct store instruction of the
            // accumulator layout touches 64 lines for 8 bytes each, and 64 of those per wave made this epilogue 17 k cycles.
            uint4 bq[MT][3];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                uint32_t pk[NT][2];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    pk[t][0] = pk_bf16(acc[m][t][0], acc[m][t][1]);
                    pk[t][1] = pk_bf16(acc[m][t][2], acc[m][t][3]);
                    if (a.relu) { pk[t][0] = relu_pk_bf16(pk[t][0], 0u); pk[t][1] = relu_pk_bf16(pk[t][1], 0u); }
                }
                bq[m][0] = make_uint4(pk[0][0], pk[0][1], pk[1][0], pk[1][1]);
                bq[m][1] = make_uint4(pk[2][0], pk[2][1], pk[3][0], pk[3][1]);
                bq[m][2] = make_uint4(pk[4 % NT][0], pk[4 % NT][1], 0u, 0u);
            }
            const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)((char*)a.dq_dst + (size_t)pg * a.dq_bytes), 0, a.dq_bytes, 0x00020000);
            const int W2 = 2 * a.Wout;
            constexpr int QPP = 128 + 8;                      // patch bytes per pixel (64 channels + 8: conflict-free 8-byte writes)
            char* const patch = smem + a.lds_patch_off + wave * (2 * 16 * QPP);   // two patches: tile m + 1 is written while tile m's lines leave
            float4 b2[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) b2[t] = *(const float4*)(a.dq_bias + t * 16 + 4 * g);
            const int posq = (pos0 + K) % RK;
            auto piece = [&](int p) -> const char* {
                int ps = posq + p / NT;
                ps = ps >= RK ? ps - RK : ps;
                return wb0 + ps * WSTEP + (p % NT) * 1024;
            };
#pragma unroll 1
            for (int ab = 0; ab < 4; ++ab) {
                // the 12 fragments of this sub-pixel: pieces 12 ab ... 12 ab + 11 (the stream behind them may still be landing)
                slow_wait(((kg + K + (12 * ab + 11) / NT) >> 1) + 1, 0);
                bf16x8 wf[4][3];
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int s3 = 0; s3 < 3; ++s3) wf[t][s3] = *(const bf16x8*)piece((ab * 4 + t) * 3 + s3);
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    f32x4 z[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) z[t] = f32x4{b2[t].x, b2[t].y, b2[t].z, b2[t].w};
#pragma unroll
                    for (int s3 = 0; s3 < 3; ++s3)
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            z[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t][s3], __builtin_bit_cast(bf16x8, bq[m][s3]), z[t], 0, 0, 0);
                    char* const pt = patch + (m & 1) * (16 * QPP);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        uint2 pk = make_uint2(pk_bf16(z[t][0], z[t][1]), pk_bf16(z[t][2], z[t][3]));
                        if (a.dq_relu) pk = make_uint2(relu_pk_bf16(pk.x, 0u), relu_pk_bf16(pk.y, 0u));
                        *(uint2*)(pt + p16 * QPP + t * 32 + g * 8) = pk;
                    }
#pragma unroll
                    for (int u = 0; u < 2; ++u) {             // 16 pixels x 8 pieces of 16 bytes: piece lane + 64 u
                        const int ii = lane + 64 * u, px = ii >> 3, c = ii & 7;
                        const char* sp = pt + px * QPP + c * 16;
                        const uint2 lo = *(const uint2*)sp, hi = *(const uint2*)(sp + 8);
                        const int y = oy0 + wave * 2 + PR(m, px), x = ox0 + PC(m, px);
                        const bool ok = y < a.Hout && x < a.Wout && c < a.dq_nch;
                        const unsigned o = ok ? (unsigned)((2 * y + (ab >> 1)) * W2 + 2 * x + (ab & 1)) * (unsigned)(a.dq_nch * 16) + (unsigned)(c * 16) : OOBS;
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, make_uint4(lo.x, lo.y, hi.x, hi.y)), rq, o, 0, 0);
                    }
                }
                // this sub-pixel's fragments are in registers: hand back the ring groups that lie wholly in front of the next one's
                asm volatile("" ::: "memory");
                if (lane == 0) flags[4 + wave] = (kg + K + (12 * (ab + 1)) / NT) >> 1;
            }
        } else {
            const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)((char*)a.dst + (size_t)pg * a.dst_bytes), 0, a.dst_bytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)(POOL ? (char*)a.pool_dst + (size_t)pg * a.pool_bytes : (char*)a.dst), 0, POOL ? a.pool_bytes : 0u, 0x00020000);
            unsigned pixoff[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int y = oy0 + wave * 2 + PR(m, p16), x = ox0 + PC(m, p16);
                pixoff[m] = (y < a.Hout && x < a.Wout) ? (unsigned)(y * a.Wout + x) * (unsigned)(CsO * 2) : OOBS;
            }
            // stores through this wave's own LDS patch (whole 128-byte lines per store instruction, see conv_mfma_kernel) where
            // the layer's LDS budget has room for one (SP_PATCH), else 8-byte lane stores
            constexpr int LST_PP = NT * 32 + 8;
            char* const patch = smem + a.lds_patch_off + wave * (2 * TW * LST_PP);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int n = t * 16 + 4 * g;
                const unsigned noff = n < CsO ? (unsigned)n * 2u : OOBS;
                float v[MT][4];
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    v[m][0] = acc[m][t][0]; v[m][1] = acc[m][t][1]; v[m][2] = acc[m][t][2]; v[m][3] = acc[m][t][3];
                    if (a.relu) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[m][r] = vmax(v[m][r], 0.0f);
                    }
                    const uint2 pk = make_uint2(pk_bf16(v[m][0], v[m][1]), pk_bf16(v[m][2], v[m][3]));
                    if constexpr (PATCH) {
                        *(uint2*)(patch + (PR(m, p16) * TW + PC(m, p16)) * LST_PP + t * 32 + g * 8) = pk;
                    } else {
                        const unsigned o = (pixoff[m] == OOBS || noff == OOBS) ? OOBS : pixoff[m] + noff;
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, pk), rd, o, 0, 0);
                    }
                }
                if constexpr (POOL) {
                    const int Wo2 = a.Wout >> 1, Ho2 = a.Hout >> 1;
#pragma unroll
                    for (int m = 0; m < 2; ++m) {             // pairs (m, m + 2): rows 2r and 2r + 1 of this wave
                        float q[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) q[r] = vmax_xor1(vmax(v[m][r], v[m + 2][r]));
                        const int y = (oy0 >> 1) + wave;
                        const int x = (ox0 >> 1) + ((m * 16 + p16) >> 1);
                        const bool ok = !(p16 & 1) && y < Ho2 && x < Wo2 && noff != OOBS;
                        const unsigned o = ok ? (unsigned)(y * Wo2 + x) * (unsigned)(CsO * 2) + noff : OOBS;
                        const uint2 pk = make_uint2(pk_bf16(q[0], q[1]), pk_bf16(q[2], q[3]));
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, pk), rp, o, 0, 0);
                    }
                }
            }
            if constexpr (PATCH) {
                constexpr int PPX = NT * 2;                   // 16-byte pieces per pixel
                constexpr int NP = 2 * TW * PPX;              // pieces of this wave's two rows
                asm volatile("" ::: "memory");                // (the patch rows are this wave's own: a wave's LDS operations execute in order)
#pragma unroll
                for (int u = 0; u < (NP + 63) / 64; ++u) {
                    const int ii = min(lane + 64 * u, NP - 1);        // (a last, partly filled trip repeats the last piece: masked below)
                    const int px = ii / PPX, c = ii - px * PPX, r = px / TW, xx = px - r * TW;
                    const char* sp = patch + px * LST_PP + c * 16;
                    const uint2 lo = *(const uint2*)sp, hi = *(const uint2*)(sp + 8);
                    const int y = oy0 + wave * 2 + r, x = ox0 + xx;
                    const bool ok = lane + 64 * u < NP && y < a.Hout && x < a.Wout && c * 16 < CsO * 2;
                    const unsigned o = ok ? (unsigned)(y * a.Wout + x) * (unsigned)(CsO * 2) + (unsigned)(c * 16) : OOBS;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, make_uint4(lo.x, lo.y, hi.x, hi.y)), rd, o, 0, 0);
                }
                asm volatile("" ::: "memory");
            }
        }
        // hand the rest of this tile's stream back (padding steps, the transposed conv's fragments)
        if (wave == 0 && i < 2) { SP_STAMP(4 + 3 * i) }
        kg += a.S;
        asm volatile("" ::: "memory");
        if (lane == 0) flags[4 + wave] = kg >> 1;
        pos0 = (pos0 + a.S) % RK;
    }
    if (wave == 0) { SP_STAMP(10) if (trc && lane == 0) { trc[8] = (unsigned long long)sw_cyc; trc[9] = (unsigned long long)sw_cnt; } }
#undef SP_STAMP
}

// ---------------------------------------------------------------------------------------------
// conv_sp2_kernel: conv_sp_kernel with TWO compute teams (two compute waves per SIMD) on one weight stream.
// Why: one compute wave per SIMD issues its MT + NT fragment reads, the counter look-ups and the MFMAs of a k-step from a single
// in-order stream -- 28-30 cycles per MFMA measured in conv_sp_kernel's k-loop against 16.7 for the MFMAs alone (tools/sp_trace.py;
// tools/microtests/kloop.hip: 22-25 for the bare loop, 18 with a second wave on the SIMD whose MFMAs fill the first one's read
// issue).  Splitting a tile's pixels or output channels over two waves raises the fragment reads per MFMA (built and removed in
// round 4: the LDS array was the limit); splitting K changes the summation order.  So the second team takes ANOTHER TILE:
//   * waves 0-3 = team 0, waves 4-7 = team 1 (one wave of each per SIMD); tile pair p of the workgroup = tiles 2p (team 0) and
//     2p + 1 (team 1); each wave keeps conv_sp_kernel's two rows x TWK pixels x NT cout tiles, k order, start value: the same bits;
//   * ONE weight ring feeds both teams: a ring group is free once all eight compute waves are past it, so the teams stay within a
//     ring's depth of each other (they read the same weight fragments at about the same time -- per pixel the weight stream from
//     L2 and its LDS-DMA writes are halved);
//   * the halo tiles go through a pool of FOUR block slots (one channel block each): the blocks of a pair are loaded in the order
//     (block 0, team 0), (block 0, team 1), (block 1, team 0) ... -- index jb, slot jb % 4 -- so each team has its current block
//     and the next one resident; a slot is free once the four waves of ITS team are past that block (tdone counts a team's own blocks);
//   * waves 8-9 stream the weights, waves 10-11 stage the blocks (two block loads in flight, published behind counted vmcnt waits);
//   * the epilogue goes through a 16-pixel LDS patch of the wave (whole 128-byte lines per store instruction), one pixel tile at a time.
// A launch with an odd tile count gives the last pair's team 1 a virtual tile: zero-filled loads, no stores.
// Counter words (ints at lds_flag_off): [0,1] ready  [2,3] tready  [4..11] done  [12..19] tdone.
// ---------------------------------------------------------------------------------------------
constexpr int SP2_NSLOT = 4;
// Bank conflicts of the pixel fragment reads.  A ds_read_b128 is served in four groups of sixteen lanes (MI355X_MICROARCH.md, LDS); with a
// pixel pitch of four slots (64 bytes: the dense block of a 32-channel slab) and lane group g reading chunk g, lanes p and p + 4 of a
// group fall on the same banks: every pixel fragment read takes the LDS array 8 cycles instead of 4 (tools/lds_conflicts.py; the same
// for pitches of 3, 5 and 7 slots, not for 6 -- conv_mfma_kernel's choice, which costs half as many slots again).  Here a four-chunk block
// keeps its dense pitch and is stored SWIZZLED: chunk c of the pixel at halo column x sits in slot c ^ 2 ((x >> 2) & 1) of that pixel -- the
// tile loaders permute the SOURCE chunk of each LDS-DMA lane (the destination stays linear), and the k-chunk table gives every lane
// the offset for ITS column parity: [k-step][lane group g][p & 7] instead of [k-step][g] (the parity of column p + kx is that of (p & 7) +
// kx: sixteen columns further on it repeats).  Conflict-free by the bank model for every tile alignment.  Five-chunk blocks (conv5)
// keep the pair order the host's bank model picks at their own pitch.
template <int NT, int SG, int FL, int TWK>
__global__ __launch_bounds__(768) __attribute__((amdgpu_waves_per_eu(3, 3))) void conv_sp2_kernel(SConv a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(TWK == 32 || TWK == 24, "tile rows of two or one and a half 16-pixel MFMA tiles");
    static_assert((FL & SP_DQ) == 0, "the fused transposed conv stays with conv_sp_kernel");
    constexpr int TW = TWK;
    constexpr int MT = TWK == 32 ? 4 : 3, TH = 8, KS = 5, THH = TH + KS - 1, TWH = TW + KS - 1, PS2 = SG * 16;
    constexpr int ROWP = sp_row_pitch(SG, TWK);
    constexpr int WSTEP = NT * 1024;
    constexpr bool POOL = (FL & SP_POOL) != 0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* const ring = smem + a.lds_ring_off;
    typedef __attribute__((address_space(3))) int lds_int;
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(3))) i32x4* lds_i32x4p;
    lds_int* const flags = (lds_int*)(smem + a.lds_flag_off);
    const lds_i32x4p flagsv = (lds_i32x4p)(smem + a.lds_flag_off);
    const int tiles_x = (a.Wout + TW - 1) / TW;
    const int npairs = (a.ntiles + 1) >> 1;
    const int n_my = (npairs - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // tile pairs of this workgroup (>= 1)
    auto xcd_pair = [&](int p) {
        if (a.xq < 0) return p;
        const int x = p & 7, j = p >> 3;
        return x * a.xq + min(x, a.xr) + j;
    };
    // tile of pair i of this workgroup and team T; ok = false: the virtual tile behind an odd tile count
    auto origin = [&](int i, int T, int& oy, int& ox, int& pg, bool& ok) {
        const int t = 2 * xcd_pair((int)blockIdx.x + i * (int)gridDim.x) + T;
        ok = t < a.ntiles;
        const int tc = ok ? t : 0;
        pg = tc / a.tiles_per_page;
        const int tl = tc - pg * a.tiles_per_page;
        const int ty = tl / tiles_x, tx = tl - ty * tiles_x;
        oy = ty * TH; ox = tx * TW;
    };
    auto give_up = [&](int code, int need, int have, int need2 = 0, int have2 = 0) {
        if (lane == 0 && atomicCAS(a.err, 0, code) == 0) { a.err[1] = wave; a.err[2] = need; a.err[3] = have; a.err[4] = need2; a.err[5] = have2; a.err[6] = (int)blockIdx.x; a.err[7] = a.layer_id; }
    };
    const int spin_limit = (PSEG_DIAG && (a.dbg & 8)) ? 64 : SP_SPIN_LIMIT;
    // PSEG_SP_TRACE (developer aid, tools/sp2_trace.py): s_memtime stamps, 16 slots per workgroup -- [0] start; team 0's wave 0: [1] / [4] first
    // fragments of tile 0 / 1 ready, [2] / [5] k-loop end, [3] / [6] epilogue end, [7] wave end, [11] cycles in slow waits; team 1's wave 4:
    // [8] ready, [9] k-loop end, [10] epilogue end of tile 0, [12] slow-wait cycles; [13] tile loader 0 end, [14] weight loader 0 end
    unsigned long long* const trc = (PSEG_DIAG && a.trace) ? a.trace + (size_t)blockIdx.x * 16 : nullptr;
#define SP2_STAMP(i) if (trc && lane == 0) trc[i] = __builtin_amdgcn_s_memtime();
    if (wave == 0) { SP2_STAMP(0) }
    if (lane == 0) {
        if (wave < 8) { flags[4 + wave] = 0; flags[12 + wave] = 0; }
        else flags[wave - 8] = 0;
    }
    lds_barrier();

    if (PSEG_DIAG && (a.dbg & 512) && wave >= 8) {           // (diagnostic build, wrong results) no loader waves at all: everything "has landed"
        if (lane == 0) flags[wave - 8] = 0x3fffffff, flags[2 + ((wave - 8) & 1)] = 0x3fffffff;
        return;
    }
    if (wave >= 10) {
        // =============================== TILE LOADERS ===============================
        const int tw = wave - 10;
        __builtin_amdgcn_s_setprio(2);
        constexpr unsigned OOB = 0xfffffff0u;
        constexpr int ROW_SLOTS = TWH * SG, J = (ROW_SLOTS + 63) >> 6;
        constexpr int PER_BLOCK = (THH / 2) * J;              // DMA instructions of one wave per block
        static_assert(THH % 2 == 0 && 2 * PER_BLOCK < 64, "counted vmcnt waits of the tile loaders");
        const unsigned inv = 65536u / (unsigned)SG + 1u;
        const int nblocks = n_my * a.nblk * 2;
        if (tw == 0) {
            const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc((void*)a.tab, 0, (unsigned)(a.K + 2) * 128u, 0x00020000);
            for (int pc = 0; pc * 1024 < (a.K + 2) * 128; ++pc)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rt, (__attribute__((address_space(3))) void*)(smem + a.lds_tab_off + pc * 1024), 16, (unsigned)(pc * 1024 + lane * 16), 0, 0, 0);
            sp_wait_vmcnt<0>();        // (counted waits below count block loads only)
        }
        auto stage = [&](int jb) {
            if (PSEG_DIAG && (a.dbg & 2)) return;
            const int T = jb & 1, q = jb >> 1;
            const int i = q / a.nblk, b = q - i * a.nblk;
            int oy0, ox0, pg; bool ok;
            origin(i, T, oy0, ox0, pg, ok);
            const int iy0 = oy0 - a.pt, ix0 = ox0 - a.pl;
            char* const in_t = smem + (jb & (SP2_NSLOT - 1)) * a.TBLK;
            const int c0 = b * a.nc_full, nc = b == a.nblk - 1 ? a.nc_last : a.nc_full;
            const bool s1 = c0 >= a.nch0;
            const int nchs = s1 ? a.nch1 : a.nch0, cs0 = s1 ? c0 - a.nch0 : c0;
            const unsigned pbytes = s1 ? a.bytes1 : a.bytes0;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)(s1 ? a.src1 : a.src0) + (size_t)pg * pbytes), 0, pbytes, 0x00020000);
            unsigned col[J];
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int sl = j * 64 + lane;
                const int px = (int)(((unsigned)sl * inv) >> 16), cs = sl - px * SG;
                const int cc = SG == 4 ? cs ^ (((px >> 2) & 1) << 1) : cs;      // the chunk that lives in this slot (see above)
                const int ix = ix0 + px;
                col[j] = (ok && cc < nc && ix >= 0 && ix < a.Win && sl < ROW_SLOTS) ? (unsigned)(ix * nchs + cs0 + cc) * 16u : OOB;
            }
            const unsigned rowb = (unsigned)a.Win * (unsigned)(nchs * 16);
            for (int py = tw; py < THH; py += 2) {
                const int iy = iy0 + py;
                const bool rowv = iy >= 0 && iy < a.Hin;
                const unsigned rb = rowv ? (unsigned)iy * rowb : OOB;
                char* drow = in_t + py * ROWP;
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    const unsigned o = (col[j] == OOB || !rowv) ? OOB : rb + col[j];
                    if (j * 64 + lane < ROW_SLOTS)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(drow + j * 1024), 16, o, 0, 0, 0);
                }
            }
        };
        int unpub = -1;                                       // the block issued last, not yet published
        for (int jb = 0; jb < nblocks; ++jb) {
            if (jb >= SP2_NSLOT) {
                // the slot's previous block jb - 4 belongs to the same team (jb & 1); it is its block number (jb - 4) >> 1
                const int need = ((jb - SP2_NSLOT) >> 1) + 1;
                const lds_i32x4p td = (lds_i32x4p)(smem + a.lds_flag_off + 48 + (jb & 1) * 16);
                for (int it = 0;; ++it) {
                    asm volatile("" ::: "memory");
                    const i32x4 d = *td;
                    if (__builtin_amdgcn_readfirstlane(min(min(d.x, d.y), min(d.z, d.w))) >= need) break;
                    if (unpub >= 0) { sp_wait_vmcnt<0>(); if (lane == 0) flags[2 + tw] = unpub + 1; unpub = -1; }   // stalled anyway: publish what has landed
                    if (it >= spin_limit) { give_up(1, need, min(min(d.x, d.y), min(d.z, d.w)), jb); break; }
                    __builtin_amdgcn_s_sleep(2);
                }
                asm volatile("" ::: "memory");
            }
            stage(jb);
            if (unpub >= 0) { sp_wait_vmcnt<PER_BLOCK>(); if (lane == 0) flags[2 + tw] = unpub + 1; }
            unpub = jb;
            // start-up (every workgroup of the launch fetches at once: ~11 B/clk/CU): team 0's first block alone, then team 1's, the rest behind them
            if (jb <= 1) { sp_wait_vmcnt<0>(); if (lane == 0) flags[2 + tw] = jb + 1; unpub = -1; }
        }
        sp_wait_vmcnt<0>();
        if (lane == 0) flags[2 + tw] = nblocks;
        if (tw == 0) { SP2_STAMP(13) }
        return;
    }
    if (wave >= 8) {
        // =============================== WEIGHT LOADERS ===============================
        const int lw = wave - 8;
        const int gpt = a.S / SP_GK;                          // groups per tile pair
        const int total = n_my * gpt;
        const int NS = a.RK / SP_GK;                          // ring slots (groups), >= 5 (host)
        // groups of this wave's loads in flight behind the newest published one: the slowest compute wave inside group c sees c + 1 once the
        // loader is at q >= c + 1 + lag, and the ring lets it reach q = c + NS - 1 -- one slot of slack at lag = NS - 3 (a five-slot ring: two
        // groups in flight; one would make the stream latency-bound: 8 KiB per ~1 k cycles for the 8 B/clk a conv5 pair needs)
        const int lag = min(3, NS - 3);
        __builtin_amdgcn_s_setprio(2);                        // a loader issues little and what it issues is late if it waits for an issue slot
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpk, 0, (unsigned)a.S * (unsigned)(NT * 1024), 0x00020000);
        const unsigned vo = (unsigned)lane * 16u;
        int qt = 0, slot = 0;
        auto issue = [&]() {
            const unsigned so = (unsigned)(qt * (SP_GK * NT) + lw) * 1024u;
            char* dstb = ring + (slot * (SP_GK * NT) + lw) * 1024;
            if (!(PSEG_DIAG && (a.dbg & 1))) {
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(dstb + j * 2048), 16, vo, so + (unsigned)j * 2048u, 0, 0);
            }
            qt = qt + 1 == gpt ? 0 : qt + 1;
            slot = slot + 1 == NS ? 0 : slot + 1;
        };
        const int pre = min(NS, total);
        const int pre0 = min(2, pre);
        for (int q = 0; q < pre0; ++q) issue();
        sp_wait_vmcnt<0>();
        if (lane == 0) flags[lw] = pre0;
        for (int it = 0; it < spin_limit; ++it) {             // the rest behind the first tile blocks
            asm volatile("" ::: "memory");
            const int t0 = flags[2], t1 = flags[3];
            if (__builtin_amdgcn_readfirstlane(min(t0, t1)) >= 1) break;
            __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
        for (int q = pre0; q < pre; ++q) issue();
#define SP_PUB(REM)                                                                              \
        if (pre - pre0 > (REM)) { sp_wait_vmcnt<(REM) * NT>(); if (lane == 0) flags[lw] = pre - (REM); }
        SP_PUB(5) SP_PUB(4) SP_PUB(3) SP_PUB(2) SP_PUB(1) SP_PUB(0)
#undef SP_PUB
        int pub = pre;
        if (PSEG_DIAG && (a.dbg & 8)) return;
        const lds_i32x4p dn = (lds_i32x4p)(smem + a.lds_flag_off + 16);
        long long wl_poll = 0;
        for (int q = pre; q < total; ++q) {
            const int need = q - NS + 1;                      // every compute wave of both teams has finished group q - NS
            const long long tp0 = trc ? __builtin_amdgcn_s_memtime() : 0;
            for (int it = 0;; ++it) {
                asm volatile("" ::: "memory");
                const i32x4 d0 = dn[0], d1 = dn[1];
                const int mn = min(min(min(d0.x, d0.y), min(d0.z, d0.w)), min(min(d1.x, d1.y), min(d1.z, d1.w)));
                if (__builtin_amdgcn_readfirstlane(mn) >= need) break;
                // the ring is full: what has been issued lands while this wave waits -- publish it now, not `lag` groups behind the next
                // issue (the consumers look at the counters a trip ahead of their need: a group published late is a slow wait)
                if (pub < q) { sp_wait_vmcnt<0>(); pub = q; if (lane == 0) flags[lw] = pub; }
                if (it >= spin_limit) { give_up(2, need, mn, q); break; }
                __builtin_amdgcn_s_sleep(1);
            }
            asm volatile("" ::: "memory");
            if (trc) wl_poll += __builtin_amdgcn_s_memtime() - tp0;
            issue();
            if (lag == 3) sp_wait_vmcnt<3 * NT>(); else if (lag == 2) sp_wait_vmcnt<2 * NT>(); else sp_wait_vmcnt<NT>();
            if (q - lag + 1 > pub) { pub = q - lag + 1; if (lane == 0) flags[lw] = pub; }
        }
        sp_wait_vmcnt<0>();
        if (lane == 0) flags[lw] = total;
        if (lw == 0) { SP2_STAMP(14) if (trc && lane == 0) trc[10] = (unsigned long long)wl_poll; }
        return;
    }
    // =============================== COMPUTE ===============================
    const int team = wave >> 2, tw4 = wave & 3;
    const int p16 = lane & 15, g = lane >> 4;
    float4 biasr[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) biasr[t] = *(const float4*)(a.bias + t * 16 + 4 * g);
    auto PR = [](int m, int p) { return TWK == 32 ? (m >> 1) : (m < 2 ? m : (p >> 3)); };
    auto PC = [](int m, int p) { return TWK == 32 ? (m & 1) * 16 + p : (m < 2 ? p : 16 + (p & 7)); };
    // pixel tiles of this wave: TWK 32: tile m at + (m >> 1) * ROWP + (m & 1) * 16 * PS2 from the lane's base (immediates); TWK 24: tiles 0 / 1
    // likewise (rows 0 / 1, columns 0-15), tile 2 straddles the two rows (columns 16-23): a lane base of its own
    // (diagnostic build, PSEG_SP_DBG & 256, wrong results: every lane of a lane group reads the same address -- fragment reads free of bank conflicts by construction)
    const bool abl_bc = PSEG_DIAG && (a.dbg & 256);
    const char* const xbase = smem + (tw4 * 2) * ROWP + (abl_bc ? 0 : p16 * PS2);
    const char* const xbase2 = smem + (tw4 * 2 + (abl_bc ? 0 : (p16 >> 3))) * ROWP + (abl_bc ? 16 * PS2 : (16 + (p16 & 7)) * PS2);
    const char* const wb0 = ring + lane * 16;
    const char* const tb = smem + a.lds_tab_off + (g * 8 + (p16 & 7)) * 4;       // table: [k-step][g][p & 7]
    const int K = a.K, RK = a.RK;
    const int CsO = a.nch_out * 8;
    constexpr unsigned OOBS = 0xfffffff0u;
    long long sw_cyc = 0, sw_w = 0;
    auto slow_wait = [&](int needw, int needt) {
        const long long tq0 = trc ? __builtin_amdgcn_s_memtime() : 0;
        bool for_w = false;
        for (int it = 0;; ++it) {
            asm volatile("" ::: "memory");
            const int r0 = flags[0], r1 = flags[1], t0 = flags[2], t1 = flags[3];
            if (it == 0) for_w = __builtin_amdgcn_readfirstlane(min(r0, r1)) < needw;
            if (__builtin_amdgcn_readfirstlane(min(r0, r1)) >= needw && __builtin_amdgcn_readfirstlane(min(t0, t1)) >= needt) break;
            if (it >= spin_limit) { give_up(3, needw, min(r0, r1), needt, min(t0, t1)); break; }
            __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
        if (trc) { const long long dt = __builtin_amdgcn_s_memtime() - tq0; sw_cyc += dt; if (for_w) sw_w += dt; }
    };
    int kg = 0, pos0 = 0;
    for (int i = 0; i < n_my; ++i) {
        int oy0, ox0, pg; bool tile_ok;
        origin(i, team, oy0, ox0, pg, tile_ok);
        const int q0 = i * a.nblk;                            // blocks of this team before this tile
        // block b of this tile: load index jb = (q0 + b) * 2 + team, slot jb % 4 = team + 2 * ((q0 + b) & 1)
        f32x4 acc[MT][NT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{biasr[n].x, biasr[n].y, biasr[n].z, biasr[n].w};
        bf16x8 xa[MT], wa[NT], xb[MT], wbq[NT];
        bool abl_x = false, abl_w = false;    // diagnostic build (PSEG_SP_DBG 32 / 64, wrong results): fragment reads only in front of the loop
#define SP_LOAD(XF, WF, POS, OFF)                                                                 \
        {                                                                                        \
            const char* wbp_ = wb0 + (POS) * WSTEP;                                              \
            const char* xp_ = xbase + (OFF);                                                     \
            if (!(PSEG_DIAG && abl_w)) WF[0] = *(const bf16x8*)(wbp_);                            \
            if (!(PSEG_DIAG && abl_x)) {                                                         \
            _Pragma("unroll") for (int m = 0; m < MT; ++m)                                       \
                XF[m] = (TWK == 24 && m == 2) ? *(const bf16x8*)(xbase2 + (OFF))                 \
                                              : *(const bf16x8*)(xp_ + (TWK == 32 ? (m >> 1) * ROWP + (m & 1) * 16 * PS2 : m * ROWP)); \
            }                                                                                    \
            if (!(PSEG_DIAG && abl_w)) {                                                         \
            _Pragma("unroll") for (int t = 1; t < NT; ++t)                                       \
                WF[t] = *(const bf16x8*)(wbp_ + t * 1024);                                       \
            }                                                                                    \
        }
#define SP_MMA(XF, WF)                                                                           \
        _Pragma("unroll") for (int t = 0; t < NT; ++t)                                           \
            _Pragma("unroll") for (int m = 0; m < MT; ++m)                                       \
                acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WF[t], XF[m], acc[m][t], 0, 0, 0);
#define SP_INTERLEAVE                                                                            \
        _Pragma("unroll") for (int q_ = 0; q_ < MT * NT; ++q_) {                                  \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                    \
            __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);                                    \
        }
        // Loop control is branch-free except for the slow wait, the block change and the two exec-masked counter stores: per trip ~75
        // scalar / branch instructions (block trackers as while loops, clamps, two needs recomputed) left the matrix pipe idle for a
        // third of a trip -- with the fragment reads AND the DMAs switched off the loop still took 28 cycles per MFMA per SIMD (tools/
        // gpu_r05_sp2_abl.sh).  Now: the table has two rows beyond K (no clamp), the block of a step is "current or next" by one compare.
#define SP_TABX(X) (*(const int*)(tb + (X) * 128))
        const int sb_lo = team * a.TBLK, sb_hi = (team + 2) * a.TBLK;
        int sb_cur = ((q0 & 1) ? sb_hi : sb_lo), sb_next = sb_lo + sb_hi - sb_cur;
        int bcur = 0, bnd = a.nblk > 1 ? a.blk_steps : 0x7fffffff;   // block of step s; first step of block bcur + 1
        int nt_cur = q0 * 2 + team + 1;                       // tready value that says block bcur of this tile has landed (next block: + 2)
        slow_wait(((kg + min(2, K - 1)) >> 1) + 1, nt_cur + ((2 >= bnd) ? 2 : 0));
        if (wave == 0 && i < 2) { SP2_STAMP(1 + 3 * i) }
        if (wave == 4 && i < 1) { SP2_STAMP(8) }
        int pos1 = pos0 + 1 == RK ? 0 : pos0 + 1, pos2 = pos1 + 1 == RK ? 0 : pos1 + 1;
        int offa = SP_TABX(0) + sb_cur, offb = SP_TABX(1) + (1 >= bnd ? sb_next : sb_cur);
        SP_LOAD(xa, wa, pos0, offa)
        if (PSEG_DIAG) { SP_LOAD(xb, wbq, pos1, offb) abl_x = (a.dbg & 32) != 0; abl_w = (a.dbg & 64) != 0; }
        int myprio = 0;
        if (PSEG_DIAG && (a.dbg & 128)) {                     // (diagnostic build, wrong results) the bare loop: MFMAs (+ fragment reads unless 32 / 64), no control
            for (int s = 0; s < K; s += 2) {
                __builtin_amdgcn_sched_barrier(0);
                SP_LOAD(xb, wbq, pos1, offb)
                SP_MMA(xa, wa)
                SP_INTERLEAVE
                __builtin_amdgcn_sched_barrier(0);
                SP_LOAD(xa, wa, pos2, offa)
                SP_MMA(xb, wbq)
                SP_INTERLEAVE
            }
        } else
        for (int s = 0; s < K; s += 2) {                      // K is even
            __builtin_amdgcn_sched_barrier(0);
            const i32x4 fv = *flagsv;
            const int pd = flags[4 + (wave ^ 4)];             // groups the partner wave on this SIMD is past (looked at behind the trip's MFMAs)
            offa = SP_TABX(s + 2) + (s + 2 >= bnd ? sb_next : sb_cur);
            SP_LOAD(xb, wbq, pos1, offb)
            SP_MMA(xa, wa)
            SP_INTERLEAVE
            __builtin_amdgcn_sched_barrier(0);
            offb = SP_TABX(s + 3) + (s + 3 >= bnd ? sb_next : sb_cur);
            SP_LOAD(xa, wa, pos2, offa)
            SP_MMA(xb, wbq)
            SP_INTERLEAVE
            __builtin_amdgcn_sched_barrier(0);
            const int mine = ((kg + s) >> 1) + 1;
            if (lane == 0) flags[4 + wave] = mine;
            if (s + 2 >= bnd) {                               // (rare) steps s, s + 1 were the last of block bcur (or s + 1 already the next one's first)
                ++bcur;
                if (lane == 0) flags[12 + wave] = q0 + bcur;
                sb_cur = sb_next; sb_next = sb_lo + sb_hi - sb_cur;
                nt_cur += 2;
                bnd = bcur + 1 < a.nblk ? bnd + a.blk_steps : 0x7fffffff;
            }
            // needs of steps s + 3 and s + 4 (requested by the next trip): their weight group, and the block of s + 4
            const int needw = ((kg + min(s + 4, K - 1)) >> 1) + 1, needt = nt_cur + (s + 4 >= bnd ? 2 : 0);
            if (__builtin_amdgcn_readfirstlane(min(fv.x, fv.y)) < needw || __builtin_amdgcn_readfirstlane(min(fv.z, fv.w)) < needt)
                slow_wait(needw, needt);
            // The two compute waves of a SIMD (this wave and wave ^ 4) are arbitrated by priority, then age: left alone, the older team
            // sprints to the end of the ring, stalls in slow_wait, and the younger one runs on alone -- two waves per SIMD in name only.
            // The wave that is BEHIND its partner takes the priority: the teams advance group by group together.
            if (!(PSEG_DIAG && (a.dbg & 16))) {
                const int want = __builtin_amdgcn_readfirstlane(pd) < mine ? 0 : 1;
                if (want != myprio) {
                    myprio = want;
                    if (want) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
                }
            }
            asm volatile("" ::: "memory");
            pos1 = pos2 + 1 == RK ? 0 : pos2 + 1;
            pos2 = pos1 + 1 == RK ? 0 : pos1 + 1;
        }
        __builtin_amdgcn_sched_barrier(0);
#undef SP_TABX
#undef SP_INTERLEAVE
#undef SP_MMA
#undef SP_LOAD
        if (lane == 0) flags[12 + wave] = q0 + a.nblk;        // the tile's last block: read to the end
        if (wave == 0 && i < 2) { SP2_STAMP(2 + 3 * i) }
        if (wave == 4 && i < 1) { SP2_STAMP(9) }
        // the whole stream of this pair is behind this wave (padding steps included): hand it back before the epilogue
        kg += a.S;
        asm volatile("" ::: "memory");
        if (lane == 0) flags[4 + wave] = kg >> 1;
        pos0 = (pos0 + a.S) % RK;
        __builtin_amdgcn_s_setprio(0);
        // ---- epilogue from registers: one 16-pixel tile at a time through this wave's LDS patch ------------------------
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
        if (tile_ok) {
            const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)((char*)a.dst + (size_t)pg * a.dst_bytes), 0, a.dst_bytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)(POOL ? (char*)a.pool_dst + (size_t)pg * a.pool_bytes : (char*)a.dst), 0, POOL ? a.pool_bytes : 0u, 0x00020000);
            constexpr int PP = NT * 32 + 8;                   // patch bytes per pixel (+ 8: conflict-free 8-byte writes)
            constexpr int PPX = NT * 2;                       // 16-byte pieces per pixel
            constexpr int NP = 8 * PPX;                       // pieces of half a pixel tile (eight pixels: <= 64, one store per lane)
            static_assert(NP <= 64, "one piece per lane");
            char* const patch = smem + a.lds_patch_off + wave * (8 * PP);
            float pv[POOL ? NT : 1][POOL ? MT : 1][4];        // (pool: the values of the tiles, paired below)
            const int px8 = min(lane / PPX, 7), pc = lane - (lane / PPX) * PPX;   // this lane's piece of a half tile: pixel, 16-byte piece
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                uint2 pk[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    float v[4] = {acc[m][t][0], acc[m][t][1], acc[m][t][2], acc[m][t][3]};
                    if (a.relu) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = vmax(v[r], 0.0f);
                    }
                    if constexpr (POOL) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) pv[t][m][r] = v[r];
                    }
                    pk[t] = make_uint2(pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3]));
                }
                if (a.dst_bytes) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {             // pixels 8 h ... 8 h + 7 of the tile
                        if ((p16 >> 3) == h) {
#pragma unroll
                            for (int t = 0; t < NT; ++t) *(uint2*)(patch + (p16 & 7) * PP + t * 32 + g * 8) = pk[t];
                        }
                        asm volatile("" ::: "memory");        // (the patch is this wave's own: a wave's LDS operations execute in order)
                        const char* sp = patch + px8 * PP + pc * 16;
                        const uint2 lo = *(const uint2*)sp, hi = *(const uint2*)(sp + 8);
                        const int px = h * 8 + px8;
                        const int y = oy0 + tw4 * 2 + PR(m, px), x = ox0 + PC(m, px);
                        const bool ok = lane < NP && y < a.Hout && x < a.Wout && pc * 16 < CsO * 2;
                        const unsigned o = ok ? (unsigned)(y * a.Wout + x) * (unsigned)(CsO * 2) + (unsigned)(pc * 16) : OOBS;
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, make_uint4(lo.x, lo.y, hi.x, hi.y)), rd, o, 0, 0);
                        asm volatile("" ::: "memory");
                    }
                }
            }
            if constexpr (POOL) {
                const int Wo2 = a.Wout >> 1, Ho2 = a.Hout >> 1;
                const int y = (oy0 >> 1) + tw4;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int n = t * 16 + 4 * g;
                    const unsigned noff = n < CsO ? (unsigned)n * 2u : OOBS;
                    // two vertical pairs either way: TWK 32: tiles (0, 2) and (1, 3); TWK 24: tiles (0, 1), and the two lane halves of tile 2
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        float q[4];
                        int xcol; bool lane_ok;
                        if (TWK == 32) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) q[r] = vmax_xor1(vmax(pv[t][pr][r], pv[t][pr + 2][r]));
                            xcol = pr * 16 + p16; lane_ok = !(p16 & 1);
                        } else if (pr == 0) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) q[r] = vmax_xor1(vmax(pv[t][0][r], pv[t][1][r]));
                            xcol = p16; lane_ok = !(p16 & 1);
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float w = pv[t][MT - 1][r];
                                // the other row's value of the same column sits eight lanes away in the 16-lane row: DPP row_ror:8
                                const float o8 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, w), 0x128, 0xF, 0xF, false));
                                q[r] = vmax_xor1(vmax(w, o8));
                            }
                            xcol = 16 + (p16 & 7); lane_ok = !(p16 & 1) && p16 < 8;
                        }
                        const int x = (ox0 + xcol) >> 1;
                        const bool ok = lane_ok && y < Ho2 && x < Wo2 && noff != OOBS;
                        const unsigned o = ok ? (unsigned)(y * Wo2 + x) * (unsigned)(CsO * 2) + noff : OOBS;
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, make_uint2(pk_bf16(q[0], q[1]), pk_bf16(q[2], q[3]))), rp, o, 0, 0);
                    }
                }
            }
        }
        if (wave == 0 && i < 2) { SP2_STAMP(3 + 3 * i) }
    }
    if (wave == 0) { SP2_STAMP(7) if (trc && lane == 0) { trc[11] = (unsigned long long)sw_cyc; trc[15] = (unsigned long long)sw_w; } }
    if (wave == 4 && trc && lane == 0) trc[12] = (unsigned long long)sw_cyc;
#undef SP2_STAMP
}

// =============================================================================================
// host side: plans, packing, launches
// =============================================================================================
enum PlanKind { PLAN_GENERIC = 0, PLAN_CONV1 = 1, PLAN_LOGITS = 2, PLAN_UPSPLIT = 3 };

struct MfmaPlan {
    int kind = PLAN_GENERIC;
    bool pp = false;          // eligible for the persistent ping-pong kernel (conv_pp_kernel) when the page has enough tiles
    int MT = 4, NT = 2, KS = 1, stride = 1, NW = 4;
    int nblk = 1, nc_full = 1, nc_last = 1, ks_full = 1, ks_last = 1;
    int PS2 = 0, row_pitch = 0, THH = 0, TWH = 0, GK = 4, NB = 3, G = 1, lds_w_off = 0, lds_tab_off = 0, lds_bytes = 0;
    int NTtot = 0, nblocks_n = 1, CoP = 0, lds_f1_off = 0;
    int* d_tab_full = nullptr;
    int* d_tab_last = nullptr;
    uint16_t* d_wpk = nullptr;
    float* d_bias = nullptr;
    float* d_wf = nullptr;   // conv1 / logits: f32 (bf16-rounded) weights
    float* d_lut = nullptr;  // conv1: bf16-rounded x/255 table
    uint16_t* d_wpk32 = nullptr;   // conv1 (k5, <= 32 couts): 32x32x16 A fragments for conv12_ws_kernel's producers
    uint16_t* d_tail_wa = nullptr;
    uint16_t* d_tail_wb = nullptr;
    float* d_tail_bias = nullptr;
    // composed tail (deconv5 o logits as one GEMM, see tail_composed_kernel)
    uint16_t* d_tc_wA1 = nullptr;
    uint16_t* d_tc_wA2 = nullptr;
    float* d_tc_beta = nullptr;
    int tc_CP = 0, tc_nks0 = 0, tc_nkss = 0;
    // host copies of the (transformed) kernel / bias: the plan is re-packed when the canvas size
    // makes the other workgroup shape (NW) the better one
    std::vector<float> w_keep, b_keep;
    int nw_tried = 0;
    bool wg3 = false;           // dense tile + 53 KB plan: three workgroups per CU
    bool nw8_ok = false;        // an 8-wave kernel instance exists for this layer shape
    bool nw8_resident = false;  // ... and its whole weight set stays resident beside the 16-row tile
    int cmax = 4;
    uint16_t* d_dq_w = nullptr;    // deconv fused behind a conv (Op::dq_fuse): A fragments [ab][cout tile 4][k-step 3][64][8], bias[64]
    float* d_dq_bias = nullptr;
    uint16_t* d_q_w = nullptr;     // into_tail deconv: A fragments [ab][2][4][64][8] and bias[32] for tail_fused2_kernel
    float* d_q_bias = nullptr;
    uint16_t* d_t2_wD = nullptr;   // composed tail with the inner deconv: composed kernel fragments
    uint16_t* d_t2_wC = nullptr;
    bool tail2 = false;
    bool pairc2 = false;           // fused conv1+conv2 (conv12_ws_kernel): channels 16-19 of two neighbouring pixels share a k-chunk (17 k-steps instead of 19)
    float* d_skiplog = nullptr;    // skip-logits buffer [canvas pixel][skip_CP] (op.skiplog >= 0), grown with the canvas
    size_t skiplog_bytes = 0;
    int skip_CP = 0;
    UpSplit* upsplit = nullptr;   // PLAN_UPSPLIT: GEMM + gather-sum form of upsample -> k2 conv
    // conv_sp_kernel (streamed weights, persistent workgroups, loader waves): a second packing of the same layer
    bool sp = false;
    int sp_NT = 0, sp_sigma = 0, sp_fl = 0, sp_K = 0, sp_S = 0, sp_blk_steps = 0, sp_nblk = 0, sp_nc_full = 0, sp_nc_last = 0;
    int sp_row_pitch = 0, sp_TBLK = 0, sp_RK = 0, sp_ring_off = 0, sp_patch_off = 0, sp_tab_off = 0, sp_flag_off = 0, sp_lds = 0;
    // the same layer on tiles of 8 x 24 pixels (conv_sp_kernel<..., 24>): its own LDS layout and k-chunk table, the same weight stream
    struct SpNarrow { bool ok = false; int row_pitch = 0, TBLK = 0, RK = 0, ring_off = 0, patch_off = 0, tab_off = 0, flag_off = 0, lds = 0; int* d_tab = nullptr; } sp24;
    int* d_sp_tab = nullptr;
    uint16_t* d_sp_wpk = nullptr;
    // conv_sp2_kernel (two compute teams, four block slots): [0] tiles of 8 x 32, [1] of 8 x 24; block-relative k-chunk tables, the same weight stream
    struct Sp2 { bool ok = false; int TBLK = 0, RK = 0, ring_off = 0, patch_off = 0, tab_off = 0, flag_off = 0, lds = 0; int* d_tab = nullptr; } sp2[2];
};

static bool sp2_off() {
    const char* v = PSEG_KNOB("PSEG_SP2");
    return v && atoi(v) == 0;
}
static bool tracing_req(const Op& op) {
    const char* trl = PSEG_DIAG_KNOB("PSEG_SP_TRACE");
    return trl && op.layer == trl;
}

void mfma_free_op(Op& op) {
    auto* p = (MfmaPlan*)op.plan;
    if (!p) return;
    (void)hipFree(p->d_tab_full); (void)hipFree(p->d_tab_last); (void)hipFree(p->d_wpk);
    (void)hipFree(p->d_bias); (void)hipFree(p->d_wf); (void)hipFree(p->d_lut); (void)hipFree(p->d_wpk32);
    (void)hipFree(p->d_tail_wa); (void)hipFree(p->d_tail_wb); (void)hipFree(p->d_tail_bias);
    (void)hipFree(p->d_tc_wA1); (void)hipFree(p->d_tc_wA2); (void)hipFree(p->d_tc_beta);
    upsplit_free(p->upsplit);
    (void)hipFree(p->d_skiplog);
    (void)hipFree(p->d_dq_w); (void)hipFree(p->d_dq_bias);
    (void)hipFree(p->d_q_w); (void)hipFree(p->d_q_bias); (void)hipFree(p->d_t2_wD); (void)hipFree(p->d_t2_wC);
    (void)hipFree(p->d_sp_tab); (void)hipFree(p->sp24.d_tab); (void)hipFree(p->d_sp_wpk); (void)hipFree(p->sp2[0].d_tab); (void)hipFree(p->sp2[1].d_tab);
    delete p;
    op.plan = nullptr;
}

// pseg_engine_trim: the canvas-sized buffers a plan owns (the skip-logits planes: 16 B per canvas pixel and page slot)
void mfma_trim_op(Op& op) {
    auto* p = (MfmaPlan*)op.plan;
    if (!p) return;
    (void)hipFree(p->d_skiplog);
    p->d_skiplog = nullptr;
    p->skiplog_bytes = 0;
}

static int producer_of(const Engine& e, int tensor) {
    for (size_t i = 0; i < e.ops.size(); ++i)
        if (e.ops[i].dst == tensor) return (int)i;
    return -1;
}

// Fold every MaxPooling2D into the epilogue of the conv that produces its input.
int mfma_plan_graph(Engine& e) {
    // Pre-activation ReLU moved into the producer (res_unet).  A conv output whose EVERY reader is a convolution that
    // applies ReLU to its input first -- conv_block 1's output (read by conv_block 2 only), the stem's first conv, the
    // bridge, e5 -- is stored ReLU'd by the conv that produces it (after the residual add, where it has one), and the
    // readers stage it as it is: bf16(max(x, 0)) == max(bf16(x), 0), so every value the readers see is unchanged, and
    // they no longer run the in-LDS ReLU pass over their halo tile (15-17 % of conv_block 2 at every level).
    if (!PSEG_KNOB("PSEG_NO_RELU_FWD") && !PSEG_KNOB("PSEG_GENERIC")) {
        std::vector<char> ok(e.tensors.size(), 0);
        for (size_t t = 0; t < e.tensors.size(); ++t) {
            const int pi = producer_of(e, (int)t);
            ok[t] = pi >= 0 && e.ops[pi].type == OP_CONV && !e.ops[pi].transposed;
        }
        for (bool changed = true; changed;) {
            changed = false;
            for (size_t t = 0; t < e.tensors.size(); ++t) {
                if (!ok[t]) continue;
                bool good = false;
                for (auto& o : e.ops) {
                    const bool reads = o.src0 == (int)t || o.src1 == (int)t;
                    if (o.add == (int)t || (reads && (o.type != OP_CONV || !o.in_relu))) { good = false; break; }
                    if (!reads) continue;
                    good = true;
                    // the reader's ReLU covers both of its sources: the other one has to move as well
                    const int other = o.src0 == (int)t ? o.src1 : o.src0;
                    if (other >= 0 && other != (int)t && !ok[other]) { good = false; break; }
                }
                if (!good) { ok[t] = 0; changed = true; }
            }
        }
        for (size_t t = 0; t < e.tensors.size(); ++t) {
            if (!ok[t]) continue;
            e.ops[producer_of(e, (int)t)].relu = 1;
            e.tensors[t].relu_stored = true;
            for (auto& o : e.ops)
                if (o.src0 == (int)t || o.src1 == (int)t) o.in_relu = 0;
        }
    }
    // ... and where a tensor has raw readers too (a block's input: the shortcut conv reads x, conv_block 1 reads max(x, 0);
    // the encoder outputs also feed the decoder's concats), its producer stores a second, ReLU'd copy (MConv::dst2: 2 B per
    // element more HBM traffic) that the pre-activation readers stage as it is, when ALL sources of such a reader have one.
    // The second store costs the producer about three times its share of the HBM bandwidth (8-byte lane stores), so it
    // pays only for the low-resolution tensors (<= 8 elements per canvas pixel: e3, e4, the bridge, d1 -- measured on a
    // 2048x1536 page: producers +28 / +10 / +5 / +16 us, readers -105 / -76 us; e1 / e2 / d2 / d3: +253 us vs -198 us).
    if (!PSEG_KNOB("PSEG_NO_RELU_FWD") && !PSEG_KNOB("PSEG_NO_RELU_COPY") && !PSEG_KNOB("PSEG_GENERIC")) {
        auto can_copy = [&](int t) {
            const int pi = producer_of(e, t);
            if (pi < 0) return false;
            const Op& p = e.ops[pi];
            // producers that leave through conv_mfma_kernel's direct-store epilogue
            const Tensor& tt = e.tensors[t];
            return p.type == OP_CONV && !p.transposed && p.k == 3 && p.stride == 1 && !p.up0 && !p.up1 && p.Cin >= 8 &&
                   ((tt.C >> (2 * tt.s)) <= 8);
        };
        const size_t nops = e.ops.size();
        for (size_t ri = 0; ri < nops; ++ri) {
            Op& r = e.ops[ri];
            if (r.type != OP_CONV || !r.in_relu) continue;
            if (!can_copy(r.src0) || (r.src1 >= 0 && !can_copy(r.src1))) continue;
            for (int* sp : {&r.src0, &r.src1}) {
                if (*sp < 0) continue;
                Op& p = e.ops[producer_of(e, *sp)];
                if (p.relu_dst < 0) {
                    Tensor c = e.tensors[*sp];
                    c.name += "/relu";
                    c.d = nullptr;
                    c.base = nullptr;
                    c.bytes = 0;
                    c.page_bytes = 0;
                    e.tensors.push_back(c);
                    p.relu_dst = (int)e.tensors.size() - 1;
                }
                *sp = p.relu_dst;
            }
            r.in_relu = 0;
        }
    }
    // deconv (k2 s2) feeding only the logits layer -> one fused tail kernel
    if (!PSEG_KNOB("PSEG_NO_TAIL_FUSION"))
        for (size_t li = 0; li < e.ops.size(); ++li) {
            Op& lg = e.ops[li];
            if (lg.type != OP_LOGITS || e.n_classes > 16) continue;
            const int pi = producer_of(e, lg.src0);
            if (pi < 0 || e.ops[pi].type != OP_DECONV2 || e.ops[pi].Cout > 32) continue;
            int users = 0;
            for (auto& o : e.ops) users += (o.src0 == lg.src0) + (o.src1 == lg.src0) + (o.add == lg.src0);
            if (users != 1) continue;
            if (lg.src1 >= 0 && e.tensors[lg.src1].Cs > 32) continue;
            e.ops[pi].tail_logits = (int)li;
            e.tensors[e.ops[pi].dst].fused = true;
            lg.fused_away = true;
        }
    // conv (64 couts = one N block of four tiles, stride 1) feeding only the logits layer -> logits / softmax /
    // argmax in that conv's epilogue (unet): its 64-channel full-resolution output is never written
    if (!PSEG_KNOB("PSEG_NO_TAIL_FUSION") && !PSEG_KNOB("PSEG_NO_CONV_LOGITS"))
        for (size_t li = 0; li < e.ops.size(); ++li) {
            Op& lg = e.ops[li];
            if (lg.type != OP_LOGITS || lg.fused_away || e.n_classes > 16 || lg.src1 >= 0) continue;
            const int pi = producer_of(e, lg.src0);
            if (pi < 0) continue;
            Op& cv = e.ops[pi];
            if (cv.type != OP_CONV || cv.Cout != 64 || cv.stride != 1 || cv.k != 3 || cv.up0 || cv.up1 ||
                cv.transposed || cv.pool_dst >= 0) continue;
            if (cv.in_relu && cv.add < 0) continue;    // plain conv (unet) or [pre-activation +] residual add (res_unet)
            int users = 0;
            for (auto& o : e.ops) users += (o.src0 == lg.src0) + (o.src1 == lg.src0) + (o.add == lg.src0);
            if (users != 1) continue;
            cv.tail_logits = (int)li;
            e.tensors[cv.dst].fused = true;
            lg.fused_away = true;
        }
    // first layer (Cin = 1, k5, 20 couts) feeding only one k5 conv -> recomputed inside that conv
    if (!PSEG_KNOB("PSEG_NO_CONV1_FUSION") && e.in_ch == 1)
        for (size_t ci = 0; ci < e.ops.size(); ++ci) {
            Op& c1 = e.ops[ci];
            if (c1.type != OP_CONV || c1.src0 != e.input_tensor || c1.src1 >= 0 || c1.k != 5 || c1.Cout != 20 || c1.stride != 1) continue;
            int users = 0, user = -1;
            for (size_t j = 0; j < e.ops.size(); ++j) {
                const Op& o = e.ops[j];
                const int u = (o.src0 == c1.dst) + (o.src1 == c1.dst) + (o.add == c1.dst);
                users += u;
                if (u) user = (int)j;
            }
            if (users != 1) continue;
            Op& c2 = e.ops[user];
            if (c2.type != OP_CONV || c2.k != 5 || c2.stride != 1 || c2.src1 >= 0 || c2.up0 || c2.in_relu || c2.add >= 0 || c2.transposed || c2.Cout > 32) continue;
            c2.fuse1 = (int)ci;
            c1.fused_away = true;
            e.tensors[c1.dst].fused = true;
        }
    // Conv2DTranspose k2 s2 behind a k5 conv whose 80-channel output nothing else reads (fcn / fcn_skip: deconv1 -> deconv2, 1/8
    // -> 1/4 resolution): the transposed conv is pointwise in the conv's output pixels, so it runs on the conv's accumulators in
    // its epilogue (FL_DQ) -- one launch less, the 1/8-resolution tensor is neither written nor read
    if (!PSEG_KNOB("PSEG_NO_DQ") && !PSEG_KNOB("PSEG_GENERIC"))
        for (size_t di = 0; di < e.ops.size(); ++di) {
            Op& dq = e.ops[di];
            if (dq.type != OP_DECONV2 || dq.src1 >= 0 || dq.tail_logits >= 0 || dq.into_tail >= 0 || dq.fused_away || dq.Cout > 64) continue;
            const int pi = producer_of(e, dq.src0);
            if (pi < 0) continue;
            Op& pc = e.ops[pi];
            if (pc.type != OP_CONV || pc.k != 5 || pc.stride != 1 || pc.Cout != 80 || !pc.relu || pc.add >= 0 || pc.in_relu || pc.up0 || pc.up1 ||
                pc.pool_dst >= 0 || pc.fuse1 >= 0 || pc.tail_logits >= 0 || pc.skiplog >= 0 || pc.relu_dst >= 0) continue;
            int users = 0;
            for (auto& o : e.ops) users += (o.src0 == pc.dst) + (o.src1 == pc.dst) + (o.add == pc.dst);
            if (users != 1) continue;
            pc.dq_fuse = (int)di;
            dq.fused_away = true;
            e.tensors[pc.dst].fused = true;
        }
    for (auto& op : e.ops) {
        if (op.type != OP_POOL) continue;
        const int pi = producer_of(e, op.src0);
        if (pi >= 0 && e.ops[pi].type == OP_CONV && e.ops[pi].pool_dst < 0) {
            e.ops[pi].pool_dst = op.dst;
            op.fused_away = true;
        }
    }
    // conv outputs nothing reads but their fused pool (fcn_skip: conv4; fcn: conv2, conv4, conv6) are not written: the
    // launch gives the output buffer descriptor zero length, which makes the hardware drop the stores
    if (!PSEG_KNOB("PSEG_NO_POOL_ONLY"))
        for (auto& cv : e.ops) {
            if (cv.type != OP_CONV || cv.pool_dst < 0 || cv.add >= 0) continue;
            int users = 0;
            for (auto& o : e.ops) users += ((o.src0 == cv.dst) + (o.src1 == cv.dst) + (o.add == cv.dst)) * (o.fused_away && o.type == OP_POOL ? 0 : 1);
            if (users == 0) { cv.pool_only = true; e.tensors[cv.dst].fused = true; }
        }
    // skip connection into a composed tail (fcn_skip: conv2 -> logits): when the full-resolution conv output has no
    // other reader than its fused pool and the logits layer, the conv stores its logits contribution (4 or 8 floats
    // per pixel) instead of the tensor, and the tail adds it: 64 + 64 B/px of HBM traffic become 16 + 16 (32 + 32).
    if (!PSEG_KNOB("PSEG_NO_SKIPLOG") && !PSEG_KNOB("PSEG_NO_TAIL_COMPOSE") && !PSEG_KNOB("PSEG_NO_TAIL_FUSION") && !PSEG_KNOB("PSEG_GENERIC") &&
        !PSEG_KNOB("PSEG_NO_CONV1_FUSION") && e.n_classes <= 8)
        for (auto& dc : e.ops) {
            if (dc.type != OP_DECONV2 || dc.tail_logits < 0 || dc.relu) continue;
            Op& lg = e.ops[dc.tail_logits];
            if (lg.src1 < 0) continue;
            const int pi = producer_of(e, lg.src1);
            if (pi < 0) continue;
            Op& cv = e.ops[pi];
            if (cv.type != OP_CONV || cv.pool_dst < 0 || cv.fuse1 < 0 || cv.Cout > 32 || cv.relu || cv.add >= 0 || cv.tail_logits >= 0) continue;
            int users = 0;
            for (auto& o : e.ops) users += ((o.src0 == cv.dst) + (o.src1 == cv.dst) + (o.add == cv.dst)) * (o.fused_away && o.type == OP_POOL ? 0 : 1);
            if (users != 1) continue;                                   // the logits layer only
            const int nks0 = cdiv((e.tensors[dc.src0].Cs + (dc.src1 >= 0 ? e.tensors[dc.src1].Cs : 0)) / 8, 4);
            if (nks0 != 3) continue;                                    // the composed-tail instances that take the buffer
            cv.skiplog = dc.tail_logits;
            e.tensors[cv.dst].fused = true;
            // ... and the transposed conv in front of that tail (fcn_skip: deconv4, ReLU), whose output nothing else reads,
            // is computed inside the tail kernel (tail_fused2_kernel) instead of being stored and read back
            if (PSEG_KNOB("PSEG_NO_TAIL2") || dc.src1 < 0 || e.tensors[dc.src1].Cs > 64) continue;
            const int di = producer_of(e, dc.src0);
            if (di < 0) continue;
            Op& dq = e.ops[di];
            if (dq.type != OP_DECONV2 || !dq.relu || dq.Cout > 32 || dq.tail_logits >= 0 || dq.into_tail >= 0) continue;
            if ((e.tensors[dq.src0].Cs + (dq.src1 >= 0 ? e.tensors[dq.src1].Cs : 0)) / 8 > 16) continue;
            if (dq.src1 >= 0 && (e.tensors[dq.src0].Cs > 64 || e.tensors[dq.src1].Cs > 64)) continue;   // k-steps 0-1 <- source 0, 2-3 <- source 1
            int du = 0;
            for (auto& o : e.ops) du += (o.src0 == dq.dst) + (o.src1 == dq.dst) + (o.add == dq.dst);
            if (du != 1) continue;
            dq.into_tail = (int)(&dc - e.ops.data());
            dq.fused_away = true;
            e.tensors[dq.dst].fused = true;
        }
    // the same inner-deconv fusion for a tail without skips (fcn: deconv4 -> deconv5 o logits)
    if (!PSEG_KNOB("PSEG_NO_TAIL2") && !PSEG_KNOB("PSEG_NO_TAIL_COMPOSE") && !PSEG_KNOB("PSEG_NO_TAIL_FUSION") && !PSEG_KNOB("PSEG_GENERIC") && e.n_classes <= 8)
        for (auto& dc : e.ops) {
            if (dc.type != OP_DECONV2 || dc.tail_logits < 0 || dc.relu || dc.src1 >= 0 || e.ops[dc.tail_logits].src1 >= 0) continue;
            if (e.tensors[dc.src0].Cs != 32) continue;
            const int di = producer_of(e, dc.src0);
            if (di < 0) continue;
            Op& dq = e.ops[di];
            if (dq.type != OP_DECONV2 || !dq.relu || dq.Cout > 32 || dq.tail_logits >= 0 || dq.into_tail >= 0) continue;
            if ((e.tensors[dq.src0].Cs + (dq.src1 >= 0 ? e.tensors[dq.src1].Cs : 0)) / 8 > 16) continue;
            if (dq.src1 >= 0 && (e.tensors[dq.src0].Cs > 64 || e.tensors[dq.src1].Cs > 64)) continue;   // k-steps 0-1 <- source 0, 2-3 <- source 1
            int du = 0;
            for (auto& o : e.ops) du += (o.src0 == dq.dst) + (o.src1 == dq.dst) + (o.add == dq.dst);
            if (du != 1) continue;
            dq.into_tail = (int)(&dc - e.ops.data());
            dq.fused_away = true;
            e.tensors[dq.dst].fused = true;
        }
    return PSEG_OK;
}

// sigma: LDS pixel stride in 16-byte slots; sigma = 2 (mod 4) and >= nc
static int sigma_for(int nc) {
    int s = std::max(nc, 2);
    while (s % 4 != 2) ++s;
    return s;
}

struct Chunk { int tap, cc; };  // cc < 0: dummy (zero weights)

// ---- LDS bank model of the im2col fragment read (MI355X_MICROARCH.md, LDS) ---------------------
// One ds_read_b128 is served in four 16-lane groups; groups {rows 0-3,12-15 of g0 + rows 4-11 of
// g1} and {rows 4-11 of g0 + rows 0-3,12-15 of g1} involve the k-chunks of lane groups g0/g1 only
// (same for g2/g3), so the cost of a k-step is the sum over its two chunk pairs.  A pair costs 2
// LDS cycles when conflict-free, up to 4 otherwise.  Offsets are in 16-byte slots; the pixel
// stride is sigma slots.
static int pair_cost(int o0, int o1, int sigma) {
    static const int RA[8] = {0, 1, 2, 3, 12, 13, 14, 15}, RB[8] = {4, 5, 6, 7, 8, 9, 10, 11};
    int cost = 0;
    for (int grp = 0; grp < 2; ++grp) {
        int cnt[16] = {0};
        const int* r0 = grp == 0 ? RA : RB;
        const int* r1 = grp == 0 ? RB : RA;
        for (int i = 0; i < 8; ++i) cnt[((sigma * r0[i] + o0) % 16 + 16) % 16]++;
        for (int i = 0; i < 8; ++i) cnt[((sigma * r1[i] + o1) % 16 + 16) % 16]++;
        int mx = 0;
        for (int i = 0; i < 16; ++i) mx = std::max(mx, cnt[i]);
        cost += mx;
    }
    return cost;
}

// Order the k-chunks of a channel block into (g0,g1),(g2,g3) pairs that read conflict-free:
// greedy matching on the bank model.  pitch_slots = row pitch in 16-byte slots.  Returns the
// ordered chunks (padded with dummies to a multiple of 4) and the total modelled LDS cycles.
// pairc2: the last chunk of a pixel holds four channels of that pixel and four of its right neighbour (conv12_ws_kernel's
// tile): the k-chunk (tap (ky, kx), last chunk) with even kx covers taps kx and kx + 1, the odd-kx ones do not exist.
static std::vector<Chunk> pair_chunks(int KS, int nc, int sigma, int pitch_slots, int* cycles_out = nullptr, bool pairc2 = false, int plane_slots = 0) {
    std::vector<Chunk> all;
    for (int t = 0; t < KS * KS; ++t)
        for (int c = 0; c < nc; ++c) {
            if (pairc2 && c == nc - 1 && ((t % KS) & 1)) continue;
            all.push_back(Chunk{t, c});
        }
    // plane_slots > 0: stride-2 tile with de-interleaved columns -- tap kx reads plane kx & 1 at pixel slot kx >> 1
    auto slot = [&](const Chunk& c) {
        const int kx = c.tap % KS;
        return (c.tap / KS) * pitch_slots + (plane_slots ? (kx & 1) * plane_slots + (kx >> 1) * sigma : kx * sigma) + c.cc;
    };
    std::vector<char> used(all.size(), 0);
    std::vector<Chunk> out;
    int cycles = 0;
    for (size_t i = 0; i < all.size(); ++i) {
        if (used[i]) continue;
        used[i] = 1;
        int best = -1, bc = 99;
        // nearest partner first keeps taps of a k-step close (weights order is arbitrary anyway)
        for (size_t j = i + 1; j < all.size() && bc > 2; ++j)
            if (!used[j]) {
                const int c = pair_cost(slot(all[i]), slot(all[j]), sigma);
                if (c < bc) { bc = c; best = (int)j; }
            }
        out.push_back(all[i]);
        if (best >= 0) { used[best] = 1; out.push_back(all[best]); cycles += bc; }
        else { out.push_back(Chunk{0, -1}); cycles += 2; }
    }
    while (out.size() % 4) { out.push_back(Chunk{0, -1}); }
    if (cycles_out) *cycles_out = cycles;
    return out;
}

static bool is_conv1_special(const Engine& e, const Op& op, int* ks, int* cout) {
    if (op.type != OP_CONV || op.src0 != e.input_tensor || op.src1 >= 0 || e.in_ch != 1) return false;
    if (op.stride != 1 || op.up0 || op.in_relu || op.add >= 0 || op.pool_dst >= 0) return false;
    const int combos[][2] = {{5, 20}, {3, 64}, {3, 32}, {1, 32}};
    for (auto& c : combos)
        if (op.k == c[0] && op.Cout == c[1]) { *ks = c[0]; *cout = c[1]; return true; }
    return false;
}

template <typename T>
static int upload(T** d, const std::vector<T>& h) {
    if (*d) (void)hipFree(*d);
    *d = nullptr;
    PSEG_HIP(hipMalloc((void**)d, std::max<size_t>(h.size(), 1) * sizeof(T)));
    if (!h.empty()) PSEG_HIP(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return PSEG_OK;
}

int mfma_pack_op(Engine& e, Op& op, const std::vector<float>& w, const std::vector<float>& bias) {
    mfma_free_op(op);
    auto* P = new MfmaPlan();
    op.plan = P;
    P->w_keep = w;
    P->b_keep = bias;
    const Tensor& s0 = e.tensors[op.src0];
    const Tensor* s1 = op.src1 >= 0 ? &e.tensors[op.src1] : nullptr;
    const int C0 = s0.C, Cs0 = s0.Cs, C1 = s1 ? s1->C : 0, Cs1 = s1 ? s1->Cs : 0;
    const int Cin = C0 + C1, Cout = op.Cout, k = op.k;
    auto rb = [](float f) { return bf2f(f2bf(f)); };

    int ks1, co1;
    if (is_conv1_special(e, op, &ks1, &co1)) {
        P->kind = PLAN_CONV1;
        P->KS = ks1;
        std::vector<float> wf((size_t)k * k * Cout), lut(256);
        for (size_t i = 0; i < wf.size(); ++i) wf[i] = rb(w[i]);  // Cin == 1: [tap][Cout]
        for (int i = 0; i < 256; ++i) lut[i] = rb((float)i / 255.0f);
        PSEG_TRY(upload(&P->d_wf, wf));
        PSEG_TRY(upload(&P->d_lut, lut));
        // MFMA form: A fragments [k-step][cout tile][lane][8]; lane = (cout & 15, g), k-step s:
        // kernel row ky = 4s + g, element j = kernel column kx
        const int nks = (k + 3) / 4, nt = cdiv(Cout, 16);
        std::vector<uint16_t> apk((size_t)nks * nt * 64 * 8, 0);
        for (int sidx = 0; sidx < nks; ++sidx)
            for (int q = 0; q < nt; ++q)
                for (int l = 0; l < 64; ++l) {
                    const int co = q * 16 + (l & 15), ky = 4 * sidx + (l >> 4);
                    if (co >= Cout || ky >= k) continue;
                    for (int j = 0; j < k; ++j)
                        apk[(((size_t)sidx * nt + q) * 64 + l) * 8 + j] = f2bf(w[((size_t)ky * k + j) * Cout + co]);
                }
        PSEG_TRY(upload(&P->d_wpk, apk));
        std::vector<float> bb((size_t)std::max(nt * 16, 32), 0.0f);
        for (int c = 0; c < Cout; ++c) bb[c] = bias[c];
        PSEG_TRY(upload(&P->d_bias, bb));
        if (k == 5 && Cout <= 32) {
            // v_mfma_f32_32x32x16_bf16 form: A fragment of k-step s, lane l = (cout l % 32, half l / 32): kernel row 2 s + half,
            // element j = kernel column (rows / columns past the 5 x 5 kernel carry zeros)
            std::vector<uint16_t> a32((size_t)3 * 64 * 8, 0);
            for (int sidx = 0; sidx < 3; ++sidx)
                for (int l = 0; l < 64; ++l) {
                    const int co = l & 31, ky = 2 * sidx + (l >> 5);
                    if (co >= Cout || ky >= k) continue;
                    for (int j = 0; j < k; ++j) a32[((size_t)sidx * 64 + l) * 8 + j] = f2bf(w[((size_t)ky * k + j) * Cout + co]);
                }
            PSEG_TRY(upload(&P->d_wpk32, a32));
        }
        return PSEG_OK;
    }
    if (op.type == OP_LOGITS) {
        P->kind = PLAN_LOGITS;
        P->cmax = Cout <= 4 ? 4 : (Cout <= 8 ? 8 : (Cout <= 16 ? 16 : (Cout <= 32 ? 32 : 64)));
        if (Cout > PSEG_MAXC) return fail(PSEG_EUNSUPPORTED, "bf16 mode supports at most %d classes (got %d)", PSEG_MAXC, Cout);
        std::vector<float> wf((size_t)(Cs0 + Cs1) * P->cmax, 0.0f), bb(P->cmax, 0.0f);
        for (int ci = 0; ci < Cin; ++ci) {
            const int cs = ci < C0 ? ci : Cs0 + (ci - C0);
            for (int c = 0; c < Cout; ++c) wf[(size_t)cs * P->cmax + c] = rb(w[(size_t)ci * Cout + c]);
        }
        for (int c = 0; c < Cout; ++c) bb[c] = bias[c];
        PSEG_TRY(upload(&P->d_wf, wf));
        if (P->cmax > 16) {   // more classes than one MFMA tile has rows: the one-pixel-per-thread kernel (logits_bf16_kernel<32 / 64>)
            PSEG_TRY(upload(&P->d_bias, bb));
            return PSEG_OK;
        }
        // MFMA form: A fragments [k-step][lane = (class, g)][8 channels of chunk 4s + g], bias padded to 16
        const int nks = cdiv((Cs0 + Cs1) / 8, 4);
        std::vector<uint16_t> wa((size_t)nks * 64 * 8, 0);
        for (int sidx = 0; sidx < nks; ++sidx)
            for (int l = 0; l < 64; ++l) {
                const int cls = l & 15, chunk = 4 * sidx + (l >> 4);
                if (cls >= Cout) continue;
                for (int j = 0; j < 8; ++j) {
                    const int cs = chunk * 8 + j;
                    if (cs >= Cs0 + Cs1) continue;
                    const int ci = cs < Cs0 ? (cs < C0 ? cs : -1) : (cs - Cs0 < C1 ? C0 + cs - Cs0 : -1);
                    if (ci >= 0) wa[((size_t)sidx * 64 + l) * 8 + j] = f2bf(w[(size_t)ci * Cout + cls]);
                }
            }
        PSEG_TRY(upload(&P->d_wpk, wa));
        bb.resize(16, 0.0f);
        PSEG_TRY(upload(&P->d_bias, bb));
        return PSEG_OK;
    }
    if (op.type == OP_POOL) return PSEG_OK;

    // upsample -> k2 conv of the deep decoder levels: GEMM over the source pixels + gather-sum (pseg_upsplit.hip).
    // Measured at 2048x1536 (unet): 1024->512 216 -> ~75 us, 512->256 228 -> ~110 us, 256->128 246 -> ~185 us; at
    // 128->64 the two extra passes over the full-resolution tensor cost more than the direct kernel (PSEG_UPSPLIT_MIN_CIN).
    if (op.type == OP_CONV && k == 2 && op.up0 && !s1 && op.stride == 1 && !op.in_relu && op.add < 0 && op.pool_dst < 0 &&
        op.tail_logits < 0 && op.fuse1 < 0 && !op.transposed && !PSEG_KNOB("PSEG_NO_UPSPLIT") && !PSEG_KNOB("PSEG_GENERIC")) {
        // (the two-pass form lost to the direct kernel at 128 -> 64 channels on the full-resolution map; the fused form does not)
        const int min_cin = PSEG_KNOB("PSEG_UPSPLIT_TWO_PASS") ? 256 : 64;
        if (Cin >= min_cin) {
            P->kind = PLAN_UPSPLIT;
            return upsplit_create(&P->upsplit, w, bias, Cin, Cs0, Cout, e.tensors[op.dst].Cs);
        }
    }
    // ---- generic MFMA conv / deconv2 ------------------------------------------------------------
    const bool deconv = op.type == OP_DECONV2;
    const int KS = deconv ? 1 : k;
    const bool tail = deconv && op.tail_logits >= 0;
    P->CoP = tail ? 32 : round_up(Cout, 4);
    const int Ntrue = deconv ? 4 * P->CoP : Cout;
    int NTall = cdiv(Ntrue, 16);
    int NT = NTall <= 5 ? NTall : (NTall % 5 == 0 ? 5 : (NTall % 4 == 0 ? 4 : (NTall % 3 == 0 ? 3 : 4)));
    if (tail) NT = 4;   // two sub-pixels (four cout tiles) per N block: 64 accumulators per lane
    // (all four sub-pixels of a k2 s2 transposed conv in one workgroup, NT = 8, was measured: 58 vs 38 us on
    // deconv4 -- 128 accumulators spill at two waves per SIMD; two N blocks re-reading the tile from L2 win)
    P->NT = NT;
    P->nblocks_n = cdiv(NTall, NT);
    P->NTtot = P->nblocks_n * NT;
    P->MT = (NT <= 2 && !deconv) ? 8 : 4;
    // stride 2 (res_unet's encoder): the halo tile of an 8 x 32 output tile is 17 x 65 pixels -- 71 KB at 32 channels, one
    // four-wave workgroup per CU beside its weights.  Four-row tiles (9 x 65 pixels) fit twice.
    const bool deint = !deconv && op.stride == 2;
    if (deint && NT == 4 && KS == 3 && !PSEG_KNOB("PSEG_NO_S2_MT2") && !PSEG_KNOB("PSEG_GENERIC")) P->MT = 2;
    // (the fused conv1+conv2 kernel was also tried with 8-row tiles -- 136 registers, 42 KB, three workgroups per
    // CU: 182 vs 164 us; the halo recompute of conv1 grows from 1.41x to 1.69x and the weights stream twice as often)
    // (the 1/8-resolution k5 layers -- conv7, deconv1: 192 eight-row tiles for 256 CUs on a 2048x1536 page -- were tried
    // with four-row tiles (384 workgroups: 28.0 -> 26.3 and 33.3 -> 31.0 us) and with two N blocks of 3 + 2 cout tiles
    // (no change): two workgroups sharing a CU gain little over one here, not worth the extra instances)
    P->KS = KS;
    P->stride = deconv ? 1 : op.stride;
    // 8-wave workgroups (16-row tiles, the whole CU's LDS) exist for the k5 stride-1 mid-layer shapes
    P->nw8_ok = !deconv && (KS == 5 || (KS == 3 && NT == 4)) && op.stride == 1 && P->MT == 4 && (NT == 3 || NT == 4) && !op.up0 && !op.up1 &&
                !op.in_relu && op.add < 0 && op.fuse1 < 0 && (P->nblocks_n == 1 || KS == 3) && !PSEG_KNOB("PSEG_GENERIC");
    P->NW = (P->nw8_ok && op.nw_hint == 8) ? 8 : 4;
    const int TH = P->NW * (P->MT / 2);
    const int totc = (Cs0 + Cs1) / 8;
    // fused conv1 + conv2 on the wave-specialised kernel: 20 input channels = 2.5 chunks per pixel -- the half chunk pairs up
    // with the right neighbour's (the producers write the tile that way), 65 k-chunks instead of 75
    P->pairc2 = op.fuse1 >= 0 && !deconv && KS == 5 && C0 == 20 && Cs0 == 24 && !s1 && op.stride == 1 && op.pool_dst >= 0 && !op.relu &&
                op.add < 0 && !op.in_relu && Cout <= 32 && !PSEG_KNOB("PSEG_NO_WS") && !PSEG_KNOB("PSEG_NO_PAIRC2") && !PSEG_KNOB("PSEG_NO_PERSIST") &&
                !PSEG_KNOB("PSEG_GENERIC");
    P->nc_full = totc <= 5 ? totc : (KS == 1 && totc <= 16 ? totc : 4);
    P->nblk = cdiv(totc, P->nc_full);
    P->nc_last = totc - (P->nblk - 1) * P->nc_full;
    // Layout choice.  Default: sigma = 2 (mod 4) slots per pixel and an odd row pitch (every pair
    // of consecutive chunks reads conflict-free).  If the whole layer's weights then do not fit
    // beside the tile in half a CU's LDS but would with the unpadded stride sigma = nc, take the
    // dense layout and let the bank model pick the row pitch / chunk pairing (a few 2-way
    // conflicts are far cheaper than streaming the weights group by group).
    int sigma = sigma_for(P->nc_full);
    int pitch_pad = 1;   // extra 16-byte slots per tile row
    P->THH = (TH - 1) * P->stride + KS;
    P->TWH = (TW - 1) * P->stride + KS;
    {
        const int ksf = cdiv(KS * KS * P->nc_full, 4);
        auto fits = [&](int sg, int pad) {
            const int in_b = P->THH * ((P->TWH * sg + pad) * 16);
            return P->nblk == 1 && in_b + round_up(ksf, 4) * NT * 1024 + ksf * 16 + 64 <= 80 * 1024;
        };
        // (the dense tile was also tried where it is not needed for residency -- conv3 at sigma = 4: 78.7 vs 79.0 us,
        // neither the third of the DMA lanes spent on pad slots nor the read conflicts matter there)
        // Three workgroups per CU (12 waves, three per SIMD) for the NT = 3 k5 layers: their instances need 147
        // registers, so only LDS limits the occupancy.  The dense tile (sigma = chunks per pixel, row pitch chosen
        // by the bank model) plus a two-slot ring fit 53 KB.  A workgroup's life is prologue -> k-loop ->
        // epilogue with the MFMA pipe used only in the middle; a third resident workgroup fills more of the
        // gaps: conv3 78 -> 67 us, deconv3 79 -> 66 us, better than the 16-row resident variant (70 us).
        P->wg3 = P->NW == 4 && !deconv && ((KS == 5 && (NT == 3 || NT == 4) && P->nblocks_n == 1) || (KS == 3 && NT == 4) ||
                                                 (KS == 2 && NT == 4 && op.up0 && !s1)) &&
                 (P->nc_full == 4 || P->nc_full == 5) && op.stride == 1 &&
                 (!op.up0 || KS == 2 || KS == 3) && !op.up1 && ((!op.in_relu && op.add < 0) || (KS == 3)) && op.fuse1 < 0 &&
                 !PSEG_KNOB("PSEG_GENERIC");
        // (stride 2: always the dense tile -- its columns are de-interleaved by parity at staging time, which makes the fragment
        // reads those of a stride-1 layer, and the 4.3 input pixels per output pixel are the layer's LDS and DMA bill)
        // conv_pp_kernel (conv3 / conv4: resident weights, two tile buffers).  (A 64-byte pixel pitch -- conv3's four chunks -- cannot be
        // read without 2-way bank conflicts: the eight lanes of a 16-lane read group that share a chunk cover four distinct 16-byte
        // windows.  An 80-byte pitch removes them and was measured in this kernel: 52.1 vs 52.0 us, so the dense tile stays.)
        P->pp = P->wg3 && KS == 5 && NT == 3 && P->nblocks_n == 1 && op.Cout <= 40 && !op.transposed && op.src1 < 0 && !PSEG_KNOB("PSEG_NO_PP");
        if (P->wg3 || (deint && P->nc_full >= 2) || (P->nblk == 1 && !fits(sigma, 1) && P->nc_full < sigma && fits(P->nc_full, 16))) {
            sigma = P->nc_full;
            int best_cyc = 1 << 30;
            for (int pad = 0; pad < 16; ++pad) {
                int cyc = 0;
                (void)pair_chunks(KS, P->nc_full, sigma, P->TWH * sigma + pad, &cyc, P->pairc2, deint ? ((P->TWH + 1) / 2) * sigma : 0);
                if (cyc < best_cyc) { best_cyc = cyc; pitch_pad = pad; }
            }
        }
    }
    if (P->pairc2) {
        if (sigma != 3) return fail(PSEG_EUNSUPPORTED, "paired half chunks need the dense 48-byte pixel (sigma 3, got %d)", sigma);
        pitch_pad = (WS_ROWP - P->TWH * sigma * 16) / 16;     // the row pitch conv12_ws_kernel is compiled for
    }
    if (P->NW != 4 && sigma != 6) return fail(PSEG_EUNSUPPORTED, "6/8-wave plan needs the sigma = 6 tile (got %d)", sigma);
    P->PS2 = sigma * 16;
    P->row_pitch = (P->TWH * sigma + pitch_pad) * 16;
    const int plane_slots = deint ? ((P->TWH + 1) / 2) * sigma : 0;   // stride 2: first slot of the odd-column plane of a tile row
    const auto ord_full = pair_chunks(KS, P->nc_full, sigma, P->row_pitch / 16, nullptr, P->pairc2, plane_slots);
    const auto ord_last = pair_chunks(KS, P->nc_last, sigma, P->row_pitch / 16, nullptr, P->pairc2, plane_slots);
    P->ks_full = (int)ord_full.size() / 4;
    P->ks_last = (int)ord_last.size() / 4;
    if ((int)ord_full.size() > MAX_TAB) return fail(PSEG_EUNSUPPORTED, "k-chunk table too large");
    auto mk_tab = [&](const std::vector<Chunk>& ord) {
        std::vector<int> t(ord.size());
        for (size_t i = 0; i < ord.size(); ++i) {
            const Chunk c = ord[i];
            if (c.cc < 0) { t[i] = 0; continue; }  // dummy: finite data, zero weights
            const int kx = c.tap % KS;
            t[i] = (c.tap / KS) * P->row_pitch + (plane_slots ? ((kx & 1) * plane_slots + (kx >> 1) * sigma) * 16 : kx * P->PS2) + c.cc * 16;
        }
        return t;
    };
    PSEG_TRY(upload(&P->d_tab_full, mk_tab(ord_full)));
    PSEG_TRY(upload(&P->d_tab_last, mk_tab(ord_last)));
    // LDS budget: input tile + a ring of NB weight groups (GK k-steps each) + table; aim at two
    // workgroups per CU (<= 80 KiB each).  GK*NT must be a multiple of 4 so that every wave
    // issues the same number of LDS-DMA loads per group (counted vmcnt).  PSEG_GK / PSEG_NB /
    // PSEG_LDS_KB override for experiments.
    const int in_bytes = P->THH * P->row_pitch;
    const int ks_max = std::max(P->ks_full, P->ks_last);
    const int tab_bytes = ks_max * 16;
    int budget = P->NW == 8 ? 156 * 1024 : 80 * 1024;
    if (P->wg3) budget = 53 * 1024;
    auto total = [&](int gk, int nbuf) { return round_up(in_bytes, 16) + nbuf * gk * NT * 1024 + tab_bytes + 16; };
    const int gstep = P->NW != 4 ? 1 : ((NT % 4 == 0) ? 1 : (NT % 2 == 0 ? 2 : 4));
    auto best_gk = [&](int nbuf) {
        int gk = 0;
        for (int c = gstep; c <= 32 && c <= round_up(ks_max, gstep); c += gstep)
            if (total(c, nbuf) <= budget) gk = c;
        return gk;
    };
    // three ring slots (two groups in flight) when that still leaves >= 4 k-steps per group
    // inside the budget, otherwise two slots with the largest group that fits
    int NB = 3, GK = best_gk(3);
    if (GK < 4 && GK < round_up(ks_max, gstep)) { NB = 2; GK = best_gk(2); }
    if (GK == 0) { NB = 2; GK = gstep; }
    // all weights of a single-block layer resident in one slot: no ring, no group barriers
    const bool can_reside = P->nblk == 1 && total(round_up(ks_max, gstep), 1) <= budget && round_up(ks_max, gstep) <= 32;
    P->nw8_resident = P->NW == 8 && can_reside;
    if (can_reside) {
        NB = 1;
        GK = round_up(ks_max, gstep);
    }
    if (total(GK, NB) > 160 * 1024) return fail(PSEG_EUNSUPPORTED, "layer %s needs %d B of LDS", op.layer.c_str(), total(GK, NB));
    P->GK = GK;
    P->NB = NB;
    const int gpb_full = cdiv(P->ks_full, GK), gpb_last = cdiv(P->ks_last, GK);
    P->G = (P->nblk - 1) * gpb_full + gpb_last;
    P->lds_w_off = round_up(in_bytes, 16);
    P->lds_tab_off = P->lds_w_off + NB * GK * NT * 1024;
    P->lds_bytes = P->lds_tab_off + tab_bytes + 16;
    if (!deconv && P->MT == 4)   // the epilogue's store patch (plain conv instances) lives in the tile's and the ring's place
        P->lds_bytes = std::max(P->lds_bytes, P->NW * (P->MT / 2) * TW * (NT * 32 + 8));
    if (op.fuse1 >= 0) {   // two bf16 copies of the (TH+8) x 48 uint8 tile
        P->lds_f1_off = round_up(P->lds_bytes, 16);
        P->lds_bytes = P->lds_f1_off + 2 * (TH + 8) * 96;
    }

    // ---- pack weights into MFMA A-fragment order -----------------------------------------------
    // concat-storage channel cs -> true input channel (or -1 for pad)
    auto true_ci = [&](int cs) {
        if (cs < Cs0) return cs < C0 ? cs : -1;
        const int c = cs - Cs0;
        return c < C1 ? C0 + c : -1;
    };
    // weight of (tap, ci, n): conv: w[(tap*Cin+ci)*Cout + n]; deconv: n = ab*CoP + co -> w[(ab*Cin+ci)*Cout+co]
    auto wval = [&](int tap, int ci, int n) -> float {
        if (!deconv) return n < Cout ? w[((size_t)tap * Cin + ci) * Cout + n] : 0.0f;
        const int ab = n / P->CoP, co = n % P->CoP;
        return (ab < 4 && co < Cout) ? w[((size_t)ab * Cin + ci) * Cout + co] : 0.0f;
    };
    // layout [group][N block][k-step in group][cout tile][lane][8]; every channel block is
    // zero-padded to whole groups
    std::vector<uint16_t> pk((size_t)P->G * GK * P->NTtot * 512, 0);
    for (int b = 0; b < P->nblk; ++b) {
        const bool last = b == P->nblk - 1;
        const auto& ord = last ? ord_last : ord_full;
        const int ksb = last ? P->ks_last : P->ks_full;
        const int kstep0 = b * gpb_full * GK;
        for (int s = 0; s < ksb; ++s)
            for (int t = 0; t < P->NTtot; ++t)
                for (int l = 0; l < 64; ++l) {
                    const Chunk c = ord[(size_t)s * 4 + (l >> 4)];
                    if (c.cc < 0) continue;
                    const int n = t * 16 + (l & 15);
                    const int kq = (kstep0 + s) / GK, ksg = (kstep0 + s) % GK, nbk = t / NT, tl = t % NT;
                    uint16_t* o = &pk[(((((size_t)kq * P->nblocks_n + nbk) * GK + ksg) * NT + tl) * 64 + l) * 8];
                    for (int j = 0; j < 8; ++j) {
                        if (P->pairc2 && c.cc == P->nc_full - 1) {
                            // elements 0-3: channels 16-19 under tap kx, elements 4-7: channels 16-19 under tap kx + 1 (the tile
                            // holds the right neighbour's four channels there); kx = 4 has no right tap
                            const int tap = c.tap + (j >> 2), ci = c.cc * 8 + (j & 3);
                            if ((c.tap % KS) + (j >> 2) < KS) o[j] = f2bf(wval(tap, ci, n));
                            continue;
                        }
                        const int ci = true_ci((b * P->nc_full + c.cc) * 8 + j);
                        if (ci >= 0) o[j] = f2bf(wval(c.tap, ci, n));
                    }
                }
    }
    PSEG_TRY(upload(&P->d_wpk, pk));
    std::vector<float> bb((size_t)P->NTtot * 16, 0.0f);
    for (int n = 0; n < P->NTtot * 16; ++n) {
        if (!deconv) { if (n < Cout) bb[n] = bias[n]; }
        else { const int ab = n / P->CoP, co = n % P->CoP; if (ab < 4 && co < Cout) bb[n] = bias[co]; }
    }
    PSEG_TRY(upload(&P->d_bias, bb));
    // ---- second packing for conv_sp_kernel: k5 stride-1 layers with one N block whose weights are streamed ----------
    if (!deconv && KS == 5 && op.stride == 1 && !op.up0 && !op.up1 && !op.in_relu && op.add < 0 && op.fuse1 < 0 && op.tail_logits < 0 &&
        op.skiplog < 0 && op.relu_dst < 0 && P->nblocks_n == 1 && NT >= 3 && NT <= 5 && (op.dq_fuse < 0 || NT == 5) && !P->pp &&
        !PSEG_KNOB("PSEG_NO_SP") && !PSEG_KNOB("PSEG_GENERIC")) {
        const int sg = P->nc_full;                 // dense tile: sigma = chunks per pixel of a block
        int pad = 1, best_cyc = 1 << 30;
        for (int pd = 0; pd < 16; ++pd) {
            int cyc = 0;
            (void)pair_chunks(KS, P->nc_full, sg, (TW + KS - 1) * sg + pd, &cyc);
            if (cyc < best_cyc) { best_cyc = cyc; pad = pd; }
        }
        const int rowp = ((TW + KS - 1) * sg + pad) * 16, TBLK = round_up((8 + KS - 1) * rowp, 16);
        const auto o_full = pair_chunks(KS, P->nc_full, sg, rowp / 16), o_last = pair_chunks(KS, P->nc_last, sg, rowp / 16);
        const int ksf = (int)o_full.size() / 4, ksl = (int)o_last.size() / 4;
        const int K0 = (P->nblk - 1) * ksf + ksl, K = round_up(K0, 2);
        const bool dq = op.dq_fuse >= 0;
        const int S = round_up(K + (dq ? cdiv(SP_DQ_PIECES, NT) : 0), SP_GK);
        const int nslot = P->nblk == 1 ? 2 : P->nblk;
        const int tiles_b = nslot * TBLK, tabb = K * 16, LDS_MAX = 160 * 1024;
        const int patchb = dq ? 4 * 2 * 16 * (128 + 8) : 4 * 2 * TW * (NT * 32 + 8);   // the transposed conv's 16-pixel patches / the plain epilogue's two rows per wave
        auto ring_for = [&](bool patch) {          // largest even ring (<= 16 steps: vmcnt counts, SP_PUB) that fits
            int rk = (LDS_MAX - tiles_b - round_up(tabb, 1024) - 64 - (patch ? patchb : 0)) / (NT * 1024);
            rk = std::min(rk, 16) & ~1;
            return rk;
        };
        const int rk_min = dq ? 12 : 10;           // five ring slots of two k-steps; the transposed conv's 10 pseudo-steps sit in the ring at once
        const bool patch = dq || ring_for(true) >= rk_min;
        const int RK = ring_for(patch);
        // the kernel instances (mfma_launch_conv), a row pitch the kernel is compiled for, channel blocks that have ONE source each,
        // at most four of them (the tile loaders' counted waits), 64 output channels behind a fused transposed conv
        const bool inst = (NT == 5 && sg == 4 && (dq || patch)) || (NT == 4 && (sg == 4 || sg == 5) && patch) || (NT == 3 && sg == 4);
        const bool shape_ok = rowp == sp_row_pitch(sg) && P->nblk <= 4 && (!s1 || (Cs0 / 8) % P->nc_full == 0) &&
                              (!dq || e.tensors[e.ops[op.dq_fuse].dst].Cs == 64);
        if (RK >= rk_min && inst && shape_ok && ksf >= 2) {
            P->sp = true;
            P->sp_NT = NT; P->sp_sigma = sg; P->sp_K = K; P->sp_S = S; P->sp_blk_steps = ksf;
            P->sp_nblk = P->nblk; P->sp_nc_full = P->nc_full; P->sp_nc_last = P->nc_last;
            P->sp_row_pitch = rowp; P->sp_TBLK = TBLK; P->sp_RK = RK;
            P->sp_fl = (dq ? SP_DQ : 0) | (patch && !dq ? SP_PATCH : 0) | (op.pool_dst >= 0 ? SP_POOL : 0);
            P->sp_ring_off = tiles_b;
            P->sp_patch_off = tiles_b + RK * NT * 1024;
            P->sp_tab_off = P->sp_patch_off + (patch ? patchb : 0);
            P->sp_flag_off = P->sp_tab_off + round_up(tabb, 1024);      // (the table arrives in 1 KiB DMA pieces)
            P->sp_lds = P->sp_flag_off + 64;
            std::vector<int> tab((size_t)K * 4, 0);
            std::vector<uint16_t> spk((size_t)S * NT * 512, 0);
            for (int b = 0; b < P->nblk; ++b) {
                const bool last = b == P->nblk - 1;
                const auto& ord = last ? o_last : o_full;
                const int ksb = last ? ksl : ksf, st0 = b * ksf, slot_base = P->nblk == 1 ? 0 : b * TBLK;
                for (int st = 0; st < ksb; ++st)
                    for (int gi = 0; gi < 4; ++gi) {
                        const Chunk c = ord[(size_t)st * 4 + gi];
                        if (c.cc < 0) continue;                          // dummy: offset 0 (finite data), zero weights
                        tab[(size_t)(st0 + st) * 4 + gi] = slot_base + (c.tap / KS) * rowp + (c.tap % KS) * sg * 16 + c.cc * 16;
                        for (int t = 0; t < NT; ++t)
                            for (int p16 = 0; p16 < 16; ++p16) {
                                uint16_t* o = &spk[(((size_t)(st0 + st) * NT + t) * 64 + gi * 16 + p16) * 8];
                                for (int j = 0; j < 8; ++j) {
                                    const int ci = true_ci((b * P->nc_full + c.cc) * 8 + j);
                                    if (ci >= 0) o[j] = f2bf(wval(c.tap, ci, t * 16 + p16));
                                }
                            }
                    }
            }
            PSEG_TRY(upload(&P->d_sp_tab, tab));
            PSEG_TRY(upload(&P->d_sp_wpk, spk));
            // the narrow tile (8 x 24, instances for the 80-channel layers): same chunk ORDER (same products in the same order: same
            // bits), offsets for its own row pitch, a ring as deep as its smaller tile allows
            if (NT == 5 && sg == 4 && op.pool_dst < 0 && !PSEG_KNOB("PSEG_NO_SP24")) {
                auto& N = P->sp24;
                const int TWN = 24, rowpn = sp_row_pitch(sg, TWN), TBLKn = round_up((8 + KS - 1) * rowpn, 16), tiles_n = nslot * TBLKn;
                const int patchn = dq ? patchb : 4 * 2 * TWN * (NT * 32 + 8);
                const int rkn = std::min((LDS_MAX - tiles_n - round_up(tabb, 1024) - 64 - patchn) / (NT * 1024), 16) & ~1;
                if (rkn >= rk_min) {
                    N.ok = true; N.row_pitch = rowpn; N.TBLK = TBLKn; N.RK = rkn;
                    N.ring_off = tiles_n; N.patch_off = tiles_n + rkn * NT * 1024; N.tab_off = N.patch_off + patchn;
                    N.flag_off = N.tab_off + round_up(tabb, 1024); N.lds = N.flag_off + 64;
                    std::vector<int> tabn((size_t)K * 4, 0);
                    for (int b = 0; b < P->nblk; ++b) {
                        const bool last = b == P->nblk - 1;
                        const auto& ord = last ? o_last : o_full;
                        const int ksb = last ? ksl : ksf, st0 = b * ksf, slot_base = P->nblk == 1 ? 0 : b * TBLKn;
                        for (int st = 0; st < ksb; ++st)
                            for (int gi = 0; gi < 4; ++gi) {
                                const Chunk c = ord[(size_t)st * 4 + gi];
                                if (c.cc >= 0) tabn[(size_t)(st0 + st) * 4 + gi] = slot_base + (c.tap / KS) * rowpn + (c.tap % KS) * sg * 16 + c.cc * 16;
                            }
                    }
                    PSEG_TRY(upload(&N.d_tab, tabn));
                }
            }
            // conv_sp2_kernel: two compute teams on a pool of four block slots (NT 3 / 4 layers; the fused transposed conv stays above)
            if (!dq && NT <= 4 && !sp2_off()) {
                for (int v = 0; v < 2; ++v) {
                    auto& Q = P->sp2[v];
                    const int twv = v == 0 ? 32 : 24, rowpv = sp_row_pitch(sg, twv), TBLKv = round_up((8 + KS - 1) * rowpv, 16);
                    const int patchv = 8 * 8 * (NT * 32 + 8), tabv = round_up((K + 2) * 128, 1024);     // table: [k-step][lane group][column & 7]
                    const int rkv = std::min((LDS_MAX - SP2_NSLOT * TBLKv - patchv - tabv - 128) / (NT * 1024), 16) & ~1;
                    if (rkv < 10) continue;
                    Q.ok = true; Q.TBLK = TBLKv; Q.RK = rkv;
                    Q.ring_off = SP2_NSLOT * TBLKv; Q.patch_off = Q.ring_off + rkv * NT * 1024; Q.tab_off = Q.patch_off + patchv;
                    Q.flag_off = Q.tab_off + tabv; Q.lds = Q.flag_off + 128;
                    std::vector<int> tabq((size_t)(K + 2) * 32, 0);       // (two rows beyond K: the loop looks its offsets up two steps ahead, unclamped)
                    for (int b = 0; b < P->nblk; ++b) {
                        const bool last = b == P->nblk - 1;
                        const auto& ord = last ? o_last : o_full;
                        const int ksb = last ? ksl : ksf, st0 = b * ksf;
                        for (int st = 0; st < ksb; ++st)
                            for (int gi = 0; gi < 4; ++gi) {
                                const Chunk c = ord[(size_t)st * 4 + gi];
                                if (c.cc < 0) continue;
                                for (int p7 = 0; p7 < 8; ++p7) {
                                    // four-chunk blocks are stored swizzled: chunk c of halo column x in slot c ^ 2 ((x >> 2) & 1); x = p + kx (+ 16)
                                    const int slot = sg == 4 ? c.cc ^ ((((p7 + c.tap % KS) >> 2) & 1) << 1) : c.cc;
                                    tabq[((size_t)(st0 + st) * 4 + gi) * 8 + p7] = (c.tap / KS) * rowpv + (c.tap % KS) * sg * 16 + slot * 16;   // inside the block's slot
                                }
                            }
                    }
                    for (int r = K; r < K + 2; ++r) std::copy(tabq.begin() + (size_t)(K - 1) * 32, tabq.begin() + (size_t)K * 32, tabq.begin() + (size_t)r * 32);
                    PSEG_TRY(upload(&Q.d_tab, tabq));
                }
            }
            if (!e.d_sp_err) {   // the engine's give-up record of conv_sp_kernel (engine_status)
                PSEG_HIP(hipMalloc((void**)&e.d_sp_err, 32));
                PSEG_HIP(hipMemset(e.d_sp_err, 0, 32));
                PSEG_HIP(hipHostMalloc((void**)&e.h_sp_err, 32, hipHostMallocDefault));
            }
            if (PSEG_KNOB("PSEG_LOG_SP"))
                fprintf(stderr, "[pseg] conv_sp plan %s: NT %d sigma %d fl %d nblk %d (nc %d / %d) K %d S %d blk_steps %d row_pitch %d TBLK %d RK %d lds %d\n", op.layer.c_str(),
                        NT, sg, P->sp_fl, P->nblk, P->nc_full, P->nc_last, K, S, ksf, rowp, TBLK, RK, P->sp_lds);
        }
    }
    if (deconv && op.into_tail >= 0) {
        // this transposed conv runs inside the composed tail: A fragments [ab][cout tile 2][k-step 4][lane = (cout & 15, g)][8],
        // element j <-> storage channel (4s + g) * 8 + j of the concatenated quarter-resolution sources
        std::vector<uint16_t> wq((size_t)4 * 2 * 4 * 64 * 8, 0);
        for (int ab = 0; ab < 4; ++ab)
            for (int t = 0; t < 2; ++t)
                for (int sidx = 0; sidx < 4; ++sidx)
                    for (int l = 0; l < 64; ++l) {
                        // one source: chunk 4s + g; two sources (each <= 8 chunks): k-steps 0-1 take source 0, k-steps 2-3 source 1, so that
                        // every fragment load of the tail kernel reads ONE tensor
                        const int co = t * 16 + (l & 15), gg = l >> 4;
                        if (co >= Cout) continue;
                        int chunk = 4 * sidx + gg;
                        if (s1) {
                            const int c2 = 4 * (sidx & 1) + gg;
                            if (c2 >= (sidx < 2 ? Cs0 : Cs1) / 8) continue;
                            chunk = sidx < 2 ? c2 : Cs0 / 8 + c2;
                        }
                        for (int j = 0; j < 8; ++j) {
                            const int cs = chunk * 8 + j;
                            if (cs >= Cs0 + Cs1) continue;
                            const int ci = true_ci(cs);
                            if (ci >= 0) wq[((((size_t)ab * 2 + t) * 4 + sidx) * 64 + l) * 8 + j] = f2bf(w[((size_t)ab * Cin + ci) * Cout + co]);
                        }
                    }
        std::vector<float> bq(32, 0.0f);
        for (int c = 0; c < Cout; ++c) bq[c] = bias[c];
        PSEG_TRY(upload(&P->d_q_w, wq));
        PSEG_TRY(upload(&P->d_q_bias, bq));
    }
    if (deconv) {
        bool fused_behind_conv = false;
        for (auto& o : e.ops) fused_behind_conv |= o.dq_fuse == (int)(&op - e.ops.data());
        if (fused_behind_conv) {
            // this transposed conv runs in the epilogue of the conv that produces its input: the B operand of k-step s is built
            // from that conv's accumulator tiles 2s and 2s + 1 -- lane (pixel, g) holds channels 16 (2s) + 4g .. + 3 in elements
            // 0-3 and 16 (2s + 1) + 4g .. + 3 in elements 4-7 -- so A fragment (ab, cout tile t, s): lane (cout 16t + l % 16,
            // g = l / 16), element j <-> input channel 16 (2s + j / 4) + 4g + j % 4
            std::vector<uint16_t> wq((size_t)4 * 4 * 3 * 64 * 8, 0);
            for (int ab = 0; ab < 4; ++ab)
                for (int t = 0; t < 4; ++t)
                    for (int sidx = 0; sidx < 3; ++sidx)
                        for (int l = 0; l < 64; ++l) {
                            const int co = t * 16 + (l & 15), gg = l >> 4;
                            if (co >= Cout) continue;
                            for (int j = 0; j < 8; ++j) {
                                const int ci = 16 * (2 * sidx + (j >> 2)) + 4 * gg + (j & 3);
                                if (ci < Cin) wq[((((size_t)ab * 4 + t) * 3 + sidx) * 64 + l) * 8 + j] = f2bf(w[((size_t)ab * Cin + ci) * Cout + co]);
                            }
                        }
            std::vector<float> bq(64, 0.0f);
            for (int c = 0; c < Cout; ++c) bq[c] = bias[c];
            PSEG_TRY(upload(&P->d_dq_w, wq));
            PSEG_TRY(upload(&P->d_dq_bias, bq));
            // conv_sp_kernel takes these 48 fragments from its weight ring: they follow the conv's k-steps in the producer's stream
            for (auto& o : e.ops) {
                auto* PC = (MfmaPlan*)o.plan;
                if (o.dq_fuse != (int)(&op - e.ops.data()) || !PC || !PC->sp) continue;
                if (!(PC->sp_fl & SP_DQ) || (size_t)(PC->sp_S - PC->sp_K) * PC->sp_NT * 512 < wq.size()) return fail(PSEG_EINVAL, "conv_sp_kernel: no room for the transposed conv's fragments");
                PSEG_HIP(hipMemcpy(PC->d_sp_wpk + (size_t)PC->sp_K * PC->sp_NT * 512, wq.data(), wq.size() * 2, hipMemcpyHostToDevice));
            }
        }
    }
    if (!deconv && op.skiplog >= 0) {
        // skip-logits fusion: the logits kernel rows of this layer's channels (they follow the deconv channels in the
        // concat, Keras (1,1,Cdec+Cout,C): w[(Cdec + co)*C + c]) as ONE A fragment: lane l = (class = l & 15, g = l >> 4),
        // element j <-> cout co = j < 4 ? 4g + j : 16 + 4g + (j - 4)
        const Op& lg = e.ops[op.skiplog];
        const std::vector<float>& lw = e.params[lg.kparam].host;
        const int C = lg.Cout, Cdec = lg.Cin - Cout;
        std::vector<uint16_t> wq(64 * 8, 0);
        for (int l = 0; l < 64; ++l) {
            const int cls = l & 15, gg = l >> 4;
            if (cls >= C) continue;
            for (int j = 0; j < 8; ++j) {
                const int co = j < 4 ? 4 * gg + j : 16 + 4 * gg + (j - 4);
                if (co < Cout) wq[l * 8 + j] = f2bf(lw[(size_t)(Cdec + co) * C + cls]);
            }
        }
        PSEG_TRY(upload(&P->d_tail_wa, wq));
        P->skip_CP = C <= 4 ? 4 : 8;
    }
    if (!deconv && op.tail_logits >= 0) {
        // conv + logits fusion: logits kernel (Keras (1,1,64,C): w[co*C + c]) in the fragment order of the two
        // epilogue MFMAs: lane l = (class = l & 15, g = l >> 4), MFMA q, element j <-> cout
        // co = j < 4 ? 16(2q) + 4g + j : 16(2q+1) + 4g + (j - 4)
        if (NT != 4 || P->nblocks_n != 1) return fail(PSEG_EUNSUPPORTED, "conv + logits fusion needs a 64-cout layer");
        const Op& lg = e.ops[op.tail_logits];
        const std::vector<float>& lw = e.params[lg.kparam].host;
        const std::vector<float>& lbias = e.params[lg.bparam].host;
        const int C = lg.Cout;
        std::vector<uint16_t> wq[2] = {std::vector<uint16_t>(64 * 8, 0), std::vector<uint16_t>(64 * 8, 0)};
        for (int q = 0; q < 2; ++q)
            for (int l = 0; l < 64; ++l) {
                const int cls = l & 15, gg = l >> 4;
                if (cls >= C) continue;
                for (int j = 0; j < 8; ++j) {
                    const int co = j < 4 ? 16 * (2 * q) + 4 * gg + j : 16 * (2 * q + 1) + 4 * gg + (j - 4);
                    if (co < Cout) wq[q][l * 8 + j] = f2bf(lw[(size_t)co * C + cls]);
                }
            }
        std::vector<float> tb(16, 0.0f);
        for (int c = 0; c < C; ++c) tb[c] = lbias[c];
        PSEG_TRY(upload(&P->d_tail_wa, wq[0]));
        PSEG_TRY(upload(&P->d_tail_wb, wq[1]));
        PSEG_TRY(upload(&P->d_tail_bias, tb));
    }
    if (tail) {
        // logits weights (Keras (1,1,Cin,C): w[ci*C + c]) in the fragment order of the fused
        // tail: lane l = (class = l & 15, g = l >> 4), element j <-> deconv channel
        // co = j < 4 ? 4g + j : 16 + 4g + (j - 4); skip channel 8g + j.
        const Op& lg = e.ops[op.tail_logits];
        const std::vector<float>& lw = e.params[lg.kparam].host;
        const std::vector<float>& lbias = e.params[lg.bparam].host;
        const int C = lg.Cout, Cd = Cout;
        const int Cskip = lg.src1 >= 0 ? e.tensors[lg.src1].C : 0;
        std::vector<uint16_t> wa(64 * 8, 0), wbk(64 * 8, 0);
        for (int l = 0; l < 64; ++l) {
            const int cls = l & 15, gg = l >> 4;
            if (cls >= C) continue;
            for (int j = 0; j < 8; ++j) {
                const int co = j < 4 ? 4 * gg + j : 16 + 4 * gg + (j - 4);
                if (co < Cd) wa[l * 8 + j] = f2bf(lw[(size_t)co * C + cls]);
                const int ch = 8 * gg + j;
                if (ch < Cskip) wbk[l * 8 + j] = f2bf(lw[(size_t)(Cd + ch) * C + cls]);
            }
        }
        std::vector<float> tb(16, 0.0f);
        for (int c = 0; c < C; ++c) tb[c] = lbias[c];
        PSEG_TRY(upload(&P->d_tail_wa, wa));
        PSEG_TRY(upload(&P->d_tail_wb, wbk));
        PSEG_TRY(upload(&P->d_tail_bias, tb));
        // ---- composed tail: M_ab = Wl_dec . Wd[ab] (from the bf16-rounded kernels, float64 products) ----
        bool skip_in_buffer = false;     // the skip producer stores its logits contribution (op.skiplog): no skip GEMM here
        if (lg.src1 >= 0) { const int sp = producer_of(e, lg.src1); skip_in_buffer = sp >= 0 && e.ops[sp].skiplog >= 0; }
        const int nks0 = cdiv((Cs0 + Cs1) / 8, 4), nkss = (Cskip > 0 && !skip_in_buffer) ? cdiv(round_up(Cskip, 8) / 8, 4) : 0;
        const int CP = C <= 4 ? 4 : (C <= 8 ? 8 : 16);
        const bool shapes_ok = (nks0 == 3 && nkss == 1) || (nks0 == 1 && nkss == 0) || (nks0 == 3 && skip_in_buffer);
        if (skip_in_buffer && (op.relu || !shapes_ok || CP > 8)) return fail(PSEG_EUNSUPPORTED, "skip-logits fusion was planned for a tail that cannot be composed");
        if (!op.relu && !PSEG_KNOB("PSEG_NO_TAIL_COMPOSE") && shapes_ok) {
            const int NTL = CP / 4;
            std::vector<double> M((size_t)4 * C * Cin, 0.0);       // [ab][cls][ci]
            for (int ab = 0; ab < 4; ++ab)
                for (int cls = 0; cls < C; ++cls)
                    for (int ci = 0; ci < Cin; ++ci) {
                        double acc = 0.0;
                        for (int co = 0; co < Cd; ++co)
                            acc += (double)rb(lw[(size_t)co * C + cls]) * (double)rb(w[((size_t)ab * Cin + ci) * Cout + co]);
                        M[((size_t)ab * C + cls) * Cin + ci] = acc;
                    }
            std::vector<uint16_t> a1((size_t)NTL * nks0 * 64 * 8, 0), a2((size_t)4 * std::max(nkss, 1) * 64 * 8, 0);
            for (int t = 0; t < NTL; ++t)
                for (int sidx = 0; sidx < nks0; ++sidx)
                    for (int l = 0; l < 64; ++l) {
                        const int row = 16 * t + (l & 15), ab = row / CP, cls = row % CP;
                        if (cls >= C) continue;
                        const int chunk = 4 * sidx + (l >> 4);
                        for (int j = 0; j < 8; ++j) {
                            const int ci = true_ci(chunk * 8 + j);
                            if (chunk * 8 + j < Cs0 + Cs1 && ci >= 0)
                                a1[(((size_t)t * nks0 + sidx) * 64 + l) * 8 + j] = f2bf((float)M[((size_t)ab * C + cls) * Cin + ci]);
                        }
                    }
            for (int ab = 0; ab < 4 && nkss; ++ab)
                for (int sidx = 0; sidx < nkss; ++sidx)
                    for (int l = 0; l < 64; ++l) {
                        const int row = 16 * (ab * CP / 16) + (l & 15);
                        if (row / CP != ab) continue;
                        const int cls = row % CP;
                        if (cls >= C) continue;
                        for (int j = 0; j < 8; ++j) {
                            const int ch = (4 * sidx + (l >> 4)) * 8 + j;
                            if (ch < Cskip) a2[(((size_t)ab * nkss + sidx) * 64 + l) * 8 + j] = f2bf(lw[(size_t)(Cd + ch) * C + cls]);
                        }
                    }
            std::vector<float> beta((size_t)NTL * 16, 0.0f);
            for (int row = 0; row < NTL * 16; ++row) {
                const int cls = row % CP;
                if (cls >= C) continue;
                double acc = (double)lbias[cls];
                for (int co = 0; co < Cd; ++co) acc += (double)rb(lw[(size_t)co * C + cls]) * (double)bias[co];
                beta[row] = (float)acc;
            }
            PSEG_TRY(upload(&P->d_tc_wA1, a1));
            PSEG_TRY(upload(&P->d_tc_wA2, a2));
            PSEG_TRY(upload(&P->d_tc_beta, beta));
            P->tc_CP = CP; P->tc_nks0 = nks0; P->tc_nkss = nkss;
            const int dqi = producer_of(e, op.src0);
            if (dqi >= 0 && e.ops[dqi].into_tail >= 0 && ((skip_in_buffer && s1) || (lg.src1 < 0 && !s1))) {
                // the same composed kernel M in the operand order of tail_fused2_kernel: the inner deconv's channels arrive as
                // two accumulator tiles (lane group g, element j <-> channel j < 4 ? 4g + j : 16 + 4g + (j - 4)), the concat
                // source as storage chunks 4s + g
                std::vector<uint16_t> wd((size_t)NTL * 64 * 8, 0), wc((size_t)NTL * 2 * 64 * 8, 0);
                for (int t = 0; t < NTL; ++t)
                    for (int l = 0; l < 64; ++l) {
                        const int row = 16 * t + (l & 15), ab = row / CP, cls = row % CP, gg = l >> 4;
                        if (cls >= C) continue;
                        for (int j = 0; j < 8; ++j) {
                            const int ch = j < 4 ? 4 * gg + j : 16 + 4 * gg + (j - 4);
                            if (ch < C0) wd[((size_t)t * 64 + l) * 8 + j] = f2bf((float)M[((size_t)ab * C + cls) * Cin + ch]);
                            for (int sidx = 0; sidx < 2; ++sidx) {
                                const int cc = (4 * sidx + gg) * 8 + j;
                                if (cc < C1) wc[(((size_t)t * 2 + sidx) * 64 + l) * 8 + j] = f2bf((float)M[((size_t)ab * C + cls) * Cin + C0 + cc]);
                            }
                        }
                    }
                PSEG_TRY(upload(&P->d_t2_wD, wd));
                PSEG_TRY(upload(&P->d_t2_wC, wc));
                P->tail2 = true;
            }
        }
    }
    return PSEG_OK;
}

template <int MT, int NT, int KS, int ST, int SG, int MODE, int FL, int NW = 4>
static int launch_inst(const MConv& a, const MfmaPlan& P, dim3 grid, hipStream_t st) {
    static bool attr_set[64] = {false};
    int dev = 0;
    PSEG_HIP(hipGetDevice(&dev));
    if (!attr_set[dev & 63]) {
        PSEG_HIP(hipFuncSetAttribute((const void*)conv_mfma_kernel<MT, NT, KS, ST, SG, MODE, FL, NW>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[dev & 63] = true;
    }
    conv_mfma_kernel<MT, NT, KS, ST, SG, MODE, FL, NW><<<grid, NW * 64, P.lds_bytes, st>>>(a);
    return PSEG_OK;
}

// Specialised instances of the hot fcn / fcn_skip layer shapes, then the runtime-generic fallback.
static int launch_generic_any2(const MConv& a, const MfmaPlan& P, dim3 grid, hipStream_t st) {
    const int mode = a.tail ? MODE_TAIL : (a.deconv ? MODE_DECONV : MODE_CONV);
    const int fl = (a.pool_dst ? FL_POOL : 0) | (a.add ? FL_ADD : 0) | (a.in_relu ? FL_INRELU : 0) |
                   (a.up0 ? FL_UP0 : 0) | (a.up1 ? FL_UP1 : 0) | (a.f1_img ? FL_FUSE1 : 0) | (a.ntiles > 0 ? FL_PERSIST : 0) |
                   ((!a.deconv && a.tail_wa && !a.skip_logits) ? FL_LOGITS : 0) | (a.skip_logits ? FL_SKIPLOG : 0) | (a.dq_w ? FL_DQ : 0);
    const int sg = a.sigma, st_ = a.stride, ks = P.KS;
    if (a.f1_img && !(P.MT == 8 && P.NT == 2 && ks == 5 && sg == 3 && mode == MODE_CONV && (fl & ~(FL_PERSIST | FL_SKIPLOG)) == (FL_POOL | FL_FUSE1)))
        return fail(PSEG_EUNSUPPORTED, "first-layer fusion has no kernel instance for this shape");
#define PSEG_TRY_INST8(MT_, NT_, KS_, ST_, SG_, MODE_, FL_)                                         \
    if (P.NW == 8 && P.MT == MT_ && P.NT == NT_ && ks == KS_ && st_ == ST_ && sg == SG_ && mode == MODE_ && fl == (FL_)) \
        return launch_inst<MT_, NT_, KS_, ST_, SG_, MODE_, (FL_), 8>(a, P, grid, st);
    PSEG_TRY_INST8(4, 3, 5, 1, 6, MODE_CONV, 0)           // conv3, deconv3 (16-row tiles)
    PSEG_TRY_INST8(4, 3, 5, 1, 6, MODE_CONV, FL_POOL)     // conv4
    PSEG_TRY_INST8(4, 4, 5, 1, 6, MODE_CONV, 0)           // conv5
    PSEG_TRY_INST8(4, 4, 5, 1, 6, MODE_CONV, FL_POOL)     // conv6
    PSEG_TRY_INST8(4, 4, 3, 1, 6, MODE_CONV, 0)           // unet k3 convs, 16-row tiles
    PSEG_TRY_INST8(4, 4, 3, 1, 6, MODE_CONV, FL_POOL)
#undef PSEG_TRY_INST8
    // (six-wave workgroups -- 12-row tiles, two per CU, three waves per SIMD -- were measured too: conv3 113 vs
    // 79 us, conv4 131 vs 93 us; 2064 workgroups on 512 slots leave a nearly empty fifth round and the smaller
    // ring doubles the group barriers)
    if (P.NW != 4) return fail(PSEG_EUNSUPPORTED, "no %d-wave kernel instance for this layer shape", P.NW);
#define PSEG_TRY_INST(MT_, NT_, KS_, ST_, SG_, MODE_, FL_)                                          \
    if (!PSEG_KNOB("PSEG_GENERIC") && P.MT == MT_ && P.NT == NT_ && ks == KS_ && st_ == ST_ && sg == SG_ && mode == MODE_ && fl == (FL_)) \
        return launch_inst<MT_, NT_, KS_, ST_, SG_, MODE_, (FL_)>(a, P, grid, st);
    PSEG_TRY_INST(8, 2, 5, 1, 3, MODE_CONV, FL_POOL | FL_FUSE1 | FL_PERSIST | FL_SKIPLOG)   // conv1 + conv2 fused, persistent, skip logits instead of the tensor
    PSEG_TRY_INST(8, 2, 5, 1, 3, MODE_CONV, FL_POOL | FL_FUSE1 | FL_SKIPLOG)
    PSEG_TRY_INST(8, 2, 5, 1, 3, MODE_CONV, FL_POOL | FL_FUSE1 | FL_PERSIST)   // conv1 + conv2 fused, persistent
    PSEG_TRY_INST(8, 2, 5, 1, 3, MODE_CONV, FL_POOL | FL_FUSE1)   // conv1 + conv2 fused
    PSEG_TRY_INST(8, 2, 5, 1, 3, MODE_CONV, FL_POOL)      // conv2 (dense tile, resident weights)
    PSEG_TRY_INST(8, 2, 5, 1, 6, MODE_CONV, FL_POOL)      // conv2 (padded tile)
    PSEG_TRY_INST(4, 3, 5, 1, 6, MODE_CONV, 0)            // conv3, deconv3
    PSEG_TRY_INST(4, 3, 5, 1, 6, MODE_CONV, FL_POOL)      // conv4
    PSEG_TRY_INST(4, 3, 5, 1, 4, MODE_CONV, 0)            // conv3, deconv3: dense 64-byte pixels, three workgroups per CU
    PSEG_TRY_INST(4, 3, 5, 1, 5, MODE_CONV, FL_POOL)      // conv4: dense 80-byte pixels, three workgroups per CU
    PSEG_TRY_INST(4, 3, 5, 1, 5, MODE_CONV, 0)
    PSEG_TRY_INST(4, 4, 5, 1, 5, MODE_CONV, 0)            // conv5: dense, three workgroups per CU (165 registers + 8 spilled)
    PSEG_TRY_INST(4, 4, 5, 1, 4, MODE_CONV, FL_POOL)      // conv6
    PSEG_TRY_INST(4, 4, 5, 1, 4, MODE_CONV, 0)
    PSEG_TRY_INST(4, 4, 5, 1, 5, MODE_CONV, FL_POOL)
    PSEG_TRY_INST(4, 4, 5, 1, 6, MODE_CONV, 0)            // conv5
    PSEG_TRY_INST(4, 4, 5, 1, 6, MODE_CONV, FL_POOL)      // conv6
    PSEG_TRY_INST(4, 5, 5, 1, 6, MODE_CONV, 0)            // conv7, deconv1
    PSEG_TRY_INST(4, 5, 5, 1, 6, MODE_CONV, FL_DQ)        // deconv1 + deconv2 (k2 s2) on its accumulators
    PSEG_TRY_INST(4, 4, 3, 1, 6, MODE_CONV, 0)            // unet: k3 convs (64..1024 channels, 32-channel blocks)
    PSEG_TRY_INST(4, 4, 3, 1, 6, MODE_CONV, FL_POOL)      // unet: k3 conv + fused pool
    PSEG_TRY_INST(4, 4, 3, 1, 4, MODE_CONV, 0)            // unet: dense tile, three workgroups per CU
    PSEG_TRY_INST(4, 4, 3, 1, 4, MODE_CONV, FL_POOL)
    PSEG_TRY_INST(4, 4, 2, 1, 4, MODE_CONV, FL_UP0)       // unet: UpSampling2D + k2 conv, dense tile
    PSEG_TRY_INST(4, 4, 3, 1, 4, MODE_CONV, FL_UP0)                   // res_unet decoder: shortcut conv on [up, skip]
    PSEG_TRY_INST(4, 4, 3, 1, 4, MODE_CONV, FL_UP0 | FL_INRELU)       // res_unet decoder: first conv of the block
    PSEG_TRY_INST(4, 4, 3, 1, 4, MODE_CONV, FL_INRELU | FL_ADD)       // res_unet: second conv + residual add
    PSEG_TRY_INST(4, 4, 3, 1, 4, MODE_CONV, FL_INRELU)                // res_unet bridge
    PSEG_TRY_INST(8, 2, 3, 1, 4, MODE_CONV, FL_INRELU | FL_ADD)       // res_unet stem: second conv (32 -> 32 at full resolution) + shortcut add
    PSEG_TRY_INST(4, 4, 3, 1, 4, MODE_CONV, FL_ADD)                   // ... the same layers reading a tensor stored after its ReLU (mfma_plan_graph)
    PSEG_TRY_INST(8, 2, 3, 1, 4, MODE_CONV, FL_ADD)
    PSEG_TRY_INST(4, 4, 3, 1, 4, MODE_CONV, FL_ADD | FL_LOGITS)
    PSEG_TRY_INST(4, 4, 3, 1, 4, MODE_CONV, FL_LOGITS)    // unet: last conv + logits + argmax
    PSEG_TRY_INST(4, 4, 3, 1, 6, MODE_CONV, FL_LOGITS)
    PSEG_TRY_INST(4, 4, 3, 1, 4, MODE_CONV, FL_INRELU | FL_ADD | FL_LOGITS)   // res_unet: last block + logits + argmax
    if (fl & FL_LOGITS) return fail(PSEG_EUNSUPPORTED, "conv + logits fusion has no kernel instance for this shape");
    if (fl & FL_SKIPLOG) return fail(PSEG_EUNSUPPORTED, "skip-logits fusion has no kernel instance for this shape");
    if (fl & FL_DQ) return fail(PSEG_EUNSUPPORTED, "the fused transposed conv has no kernel instance for this shape (MT %d NT %d k %d sigma %d flags %d)", P.MT, P.NT, ks, sg, fl);
    PSEG_TRY_INST(4, 4, 2, 1, 6, MODE_CONV, FL_UP0)       // unet: UpSampling2D + k2 conv
    // single-block stride-2 layers with resident weights as PERSISTENT workgroups (res_unet 32 -> 64): a tile of theirs is 9 k-steps --
    // tools/trace_layers.py: ring issue 1.6 k + table 2.1 k + tile stage 5.5 k + wait 1.0 k cycles around a k-loop of 3.3 k; the walk stages
    // weights and table once per workgroup: 103 -> 96 and 99 -> 85 us on one box.  (The 32 -> 32 full-resolution layer lost as a
    // persistent instance, 152 -> 157-167 us: it gives up the epilogue's LDS store patch, which lives in the resident weights' place.)
    PSEG_TRY_INST(2, 4, 3, 2, 4, MODE_CONV, FL_PERSIST)
    PSEG_TRY_INST(2, 4, 3, 2, 4, MODE_CONV, FL_INRELU | FL_PERSIST)
    PSEG_TRY_INST(2, 4, 3, 2, 4, MODE_CONV, 0)            // res_unet encoder: stride-2 shortcut conv (four-row tiles, dense de-interleaved tile)
    PSEG_TRY_INST(2, 4, 3, 2, 4, MODE_CONV, FL_INRELU)    // res_unet encoder: stride-2 first conv of the block
    PSEG_TRY_INST(4, 4, 3, 2, 4, MODE_CONV, 0)            // (eight-row tiles: PSEG_NO_S2_MT2)
    PSEG_TRY_INST(4, 4, 3, 2, 4, MODE_CONV, FL_INRELU)
    PSEG_TRY_INST(4, 5, 1, 1, 10, MODE_DECONV, 0)         // deconv2
    if (a.nb_loop == 2) { PSEG_TRY_INST(4, 4, 1, 1, 14, MODE_DECONV, 0) }   // deconv4 (fcn_skip): this instance walks two N blocks per workgroup (NBL = 2)
    PSEG_TRY_INST(4, 4, 1, 1, 6, MODE_DECONV, 0)          // deconv4 (fcn)
    if (a.nb_loop == 2) {                                 // the tail instances walk two N blocks per workgroup too
        PSEG_TRY_INST(4, 4, 1, 1, 10, MODE_TAIL, 0)       // deconv5 + logits (fcn_skip)
        PSEG_TRY_INST(4, 4, 1, 1, 6, MODE_TAIL, 0)        // deconv5 + logits (fcn)
    }
#undef PSEG_TRY_INST
    if (PSEG_KNOB("PSEG_LOG_GENERIC"))
        fprintf(stderr, "[pseg] generic instance: MT %d NT %d KS %d stride %d sigma %d mode %d flags %d\n", P.MT, P.NT, ks, P.stride, sg, mode, fl);
    if (P.MT != 4 && P.MT != 8) return fail(PSEG_EUNSUPPORTED, "no runtime-generic kernel for %d pixel tiles per wave", P.MT);
    if (P.MT == 8) {
        if (P.NT == 1) return launch_inst<8, 1, -1, -1, -1, -1, -1>(a, P, grid, st);
        return launch_inst<8, 2, -1, -1, -1, -1, -1>(a, P, grid, st);
    }
    switch (P.NT) {
        case 1: return launch_inst<4, 1, -1, -1, -1, -1, -1>(a, P, grid, st);
        case 2: return launch_inst<4, 2, -1, -1, -1, -1, -1>(a, P, grid, st);
        case 3: return launch_inst<4, 3, -1, -1, -1, -1, -1>(a, P, grid, st);
        case 4: return launch_inst<4, 4, -1, -1, -1, -1, -1>(a, P, grid, st);
        case 5: return launch_inst<4, 5, -1, -1, -1, -1, -1>(a, P, grid, st);
    }
    return fail(PSEG_EUNSUPPORTED, "no kernel instance for NT=%d (sigma %d)", P.NT, sg);
}

static int launch_generic_any(const MConv& a0, const MfmaPlan& P, dim3 grid, hipStream_t st, const char* layer) {
#if PSEG_DIAG
    const char* tr = PSEG_DIAG_KNOB("PSEG_TRACE");
    if (tr && strcmp(tr, layer) == 0) {
        MConv a = a0;
        const size_t n = (size_t)grid.x * grid.y * 12;
        PSEG_HIP(hipMalloc((void**)&a.trace, n * 8));
        PSEG_HIP(hipMemset(a.trace, 0, n * 8));
        int rc = launch_generic_any2(a, P, grid, st);
        PSEG_HIP(hipStreamSynchronize(st));
        std::vector<unsigned long long> h(n);
        PSEG_HIP(hipMemcpy(h.data(), a.trace, n * 8, hipMemcpyDeviceToHost));
        (void)hipFree(a.trace);
        std::string fn = std::string("gpurun_out/trace_") + layer + ".bin";
        if (FILE* f = fopen(fn.c_str(), "wb")) { fwrite(h.data(), 8, n, f); fclose(f); }
        return rc;
    }
#else
    (void)layer;
#endif
    return launch_generic_any2(a0, P, grid, st);
}

// compute units of the current device (cached per device: persistent kernels launch one workgroup per CU)
static int device_cus(int* dev_out = nullptr) {
    static int cache[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev_out) *dev_out = dev;
    int& n = cache[dev & 63];
    if (n == 0 && (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1)) n = 256;
    return n;
}

static void fill_common(const Engine& e, const Op& op, const MfmaPlan& P, MConv& a) {
    const Tensor& s0 = e.tensors[op.src0];
    const Tensor* s1 = op.src1 >= 0 ? &e.tensors[op.src1] : nullptr;
    a.src0 = (const uint16_t*)s0.d;
    a.src1 = s1 ? (const uint16_t*)s1->d : nullptr;
    a.nch0 = s0.Cs / 8;
    a.nch1 = s1 ? s1->Cs / 8 : 0;
    a.bytes0 = (unsigned)((size_t)e.tH(s0) * e.tW(s0) * s0.Cs * 2);
    a.bytes1 = s1 ? (unsigned)((size_t)e.tH(*s1) * e.tW(*s1) * s1->Cs * 2) : 0u;
    a.sigma = P.PS2 / 16;
    a.xq = -1; a.xr = 0;      // plain tile order unless the launcher sets the XCD bands
    a.up0 = op.up0;
    a.up1 = op.up1;
    a.Hin = e.tH(s0) << op.up0;
    a.Win = e.tW(s0) << op.up0;
    a.in_relu = op.in_relu;
    a.relu = op.relu;
    a.nblk = P.nblk; a.nc_full = P.nc_full; a.nc_last = P.nc_last; a.ks_full = P.ks_full; a.ks_last = P.ks_last;
    a.tab_full = P.d_tab_full; a.tab_last = P.d_tab_last; a.wpk = P.d_wpk; a.NTtot = P.NTtot; a.bias = P.d_bias;
    a.PS2 = P.PS2; a.row_pitch = P.row_pitch; a.THH = P.THH; a.TWH = P.TWH; a.GK = P.GK; a.NB = P.NB; a.G = P.G;
    a.lds_w_off = P.lds_w_off; a.lds_tab_off = P.lds_tab_off;
    const Tensor& d = e.tensors[op.dst];
    a.dst = (uint16_t*)d.d;
    a.dst2 = op.relu_dst >= 0 ? (uint16_t*)e.tensors[op.relu_dst].d : nullptr;
    a.nch_out = d.Cs / 8;
    a.dst_bytes = op.pool_only ? 0u : (unsigned)((size_t)e.tH(d) * e.tW(d) * d.Cs * 2);
    a.CoP = P.CoP;
    a.nb_loop = 1;
    a.nb_total = P.nblocks_n;
    a.dbg = PSEG_DIAG_KNOB("PSEG_DBG") ? atoi(PSEG_DIAG_KNOB("PSEG_DBG")) : 0;   // wrong-result ablations: diagnostic build only
    if (PSEG_KNOB("PSEG_NO_LDS_STORE")) a.dbg |= 32;   // direct 8-byte stores instead of the LDS patch (same bytes)
}

int mfma_launch_conv(Engine& e, Op& op, hipStream_t st) {
    auto* P = (MfmaPlan*)op.plan;
    if (!P) return fail(PSEG_EINVAL, "layer %s has no bf16 plan", op.layer.c_str());
    const Tensor& d = e.tensors[op.dst];
    if (op.relu_dst >= 0 && (P->kind != PLAN_GENERIC || op.pool_dst >= 0 || op.tail_logits >= 0 || op.skiplog >= 0 || op.fuse1 >= 0))
        return fail(PSEG_EUNSUPPORTED, "layer %s: the ReLU'd second output exists only in the direct-store epilogue", op.layer.c_str());
    if (P->kind == PLAN_CONV1) {
        const uint8_t* img = e.cur_img;  // raw uint8 page (x/255 and pad-to-32 are fused)
        if ((P->KS == 3 && (op.Cout == 64 || op.Cout == 32))) {
            constexpr int RB = 16;
            dim3 g(cdiv(e.Wp, 64), cdiv(e.Hp, 4 * RB));
            if (op.Cout == 64) conv1_rows_kernel<3, 64, RB><<<g, 256, 0, st>>>(img, e.H, e.W, e.Hp, e.Wp, P->d_wf, P->d_bias, (uint16_t*)d.d, op.relu);
            else conv1_rows_kernel<3, 32, RB><<<g, 256, 0, st>>>(img, e.H, e.W, e.Hp, e.Wp, P->d_wf, P->d_bias, (uint16_t*)d.d, op.relu);
            return PSEG_OK;
        }
        dim3 g1(e.Wp / 32, cdiv(e.Hp, 16));
        uint16_t* o = (uint16_t*)d.d;
        if (P->KS == 5 && op.Cout == 20) conv1_mfma_kernel<5, 20><<<g1, 256, 0, st>>>(img, e.H, e.W, e.Wp, P->d_wpk, P->d_bias, o, op.relu);
        else if (P->KS == 3 && op.Cout == 64) conv1_mfma_kernel<3, 64><<<g1, 256, 0, st>>>(img, e.H, e.W, e.Wp, P->d_wpk, P->d_bias, o, op.relu);
        else if (P->KS == 3 && op.Cout == 32) conv1_mfma_kernel<3, 32><<<g1, 256, 0, st>>>(img, e.H, e.W, e.Wp, P->d_wpk, P->d_bias, o, op.relu);
        else conv1_mfma_kernel<1, 32><<<g1, 256, 0, st>>>(img, e.H, e.W, e.Wp, P->d_wpk, P->d_bias, o, op.relu);
        return PSEG_OK;
    }
    if (P->kind == PLAN_UPSPLIT) {
        const Tensor& s0 = e.tensors[op.src0];
        return upsplit_launch(P->upsplit, (const uint16_t*)s0.d, e.tH(s0), e.tW(s0), (uint16_t*)d.d, op.relu, st);
    }
    if (P->nw8_ok) {
        // 16-row tiles when they still fill the chip (one workgroup per CU); re-pack on a change
        // Measured (MI355X, 2048x1536): 16-row tiles pay off where the layer's whole weight set then
        // stays resident in LDS (conv3: 80 -> 68 us); with a streamed ring they only tie (the halved
        // weight traffic is offset by losing the second workgroup's overlap).
        const int want = (cdiv(e.tW(d), TW) * cdiv(e.tH(d), 16) >= 224 && sigma_for(P->nc_full) == 6 && !P->wg3) ? 8 : 4;
        const char* const ev = nullptr;
        if (want != P->NW && want != P->nw_tried) {
            const std::vector<float> w = P->w_keep, b = P->b_keep;
            op.nw_hint = want;
            PSEG_TRY(mfma_pack_op(e, op, w, b));
            P = (MfmaPlan*)op.plan;
            P->nw_tried = want;      // the shape may not have an instance for `want`: do not re-pack every launch
            if (want == 8 && !P->nw8_resident && !ev) {   // ring only: keep two workgroups per CU
                op.nw_hint = 4;
                PSEG_TRY(mfma_pack_op(e, op, w, b));
                P = (MfmaPlan*)op.plan;
                P->nw8_ok = false;                        // decided for this engine
            }
        }
    }
    MConv a{};
    fill_common(e, op, *P, a);
    a.Hout = e.tH(d);
    a.Wout = e.tW(d);
    a.stride = op.stride;
    const int tot_h = std::max((a.Hout - 1) * op.stride + op.k - a.Hin, 0);
    const int tot_w = std::max((a.Wout - 1) * op.stride + op.k - a.Win, 0);
    a.pt = tot_h / 2;
    a.pl = tot_w / 2;
    if (op.transposed) { a.pt = tot_h - a.pt; a.pl = tot_w - a.pl; }
    a.pool_dst = op.pool_dst >= 0 ? (uint16_t*)e.tensors[op.pool_dst].d : nullptr;
    if (op.pool_dst >= 0) { const Tensor& pt = e.tensors[op.pool_dst]; a.pool_bytes = (unsigned)((size_t)e.tH(pt) * e.tW(pt) * pt.Cs * 2); }
    a.add = op.add >= 0 ? (const uint16_t*)e.tensors[op.add].d : nullptr;
    a.deconv = 0;
    if (op.tail_logits >= 0) {   // logits / softmax / argmax in this conv's epilogue
        const Op& lg = e.ops[op.tail_logits];
        a.tail_C = lg.Cout;
        a.H0 = e.H;
        a.W0 = e.W;
        a.tail_wa = P->d_tail_wa;
        a.tail_wb = P->d_tail_wb;
        a.tail_bias = P->d_tail_bias;
        a.out_logits = e.cur_logits;
        a.out_probs = e.cur_probs;
        a.out_labels = e.cur_labels;
        a.out_labels_u8 = e.cur_labels_u8;
    }
    if (op.skiplog >= 0) {
        const size_t need = (size_t)a.Hout * a.Wout * P->skip_CP * 4;     // one page slot
        if (need * e.pages > P->skiplog_bytes) {
            PSEG_HIP(hipDeviceSynchronize());
            (void)hipFree(P->d_skiplog);
            P->d_skiplog = nullptr; P->skiplog_bytes = 0;
            PSEG_HIP(hipMalloc((void**)&P->d_skiplog, need * e.pages));
            P->skiplog_bytes = need * e.pages;
        }
        a.skip_logits = (float*)((char*)P->d_skiplog + (size_t)e.page * need);
        a.skip_CP = P->skip_CP;
        a.tail_wa = P->d_tail_wa;
    }
    if (op.fuse1 >= 0) {
        const Op& c1 = e.ops[op.fuse1];
        auto* P1 = (MfmaPlan*)c1.plan;
        if (!P1 || P1->kind != PLAN_CONV1) return fail(PSEG_EINVAL, "fused first layer has no plan");
        a.f1_img = e.cur_img;
        a.f1_H = e.H;
        a.f1_W = e.W;
        a.f1_wpk = P1->d_wpk;
        a.f1_wpk32 = PSEG_KNOB("PSEG_NO_C32") ? nullptr : P1->d_wpk32;
        a.f1_bias = P1->d_bias;
        a.f1_relu = c1.relu;
        a.lds_f1_off = P->lds_f1_off;
    }
    if (op.dq_fuse >= 0) {
        const Op& dq = e.ops[op.dq_fuse];
        auto* PQ = (MfmaPlan*)dq.plan;
        if (!PQ || !PQ->d_dq_w) return fail(PSEG_EINVAL, "fused transposed conv: plan data missing");
        const Tensor& qd = e.tensors[dq.dst];
        a.dq_w = PQ->d_dq_w; a.dq_bias = PQ->d_dq_bias; a.dq_dst = (uint16_t*)qd.d; a.dq_nch = qd.Cs / 8; a.dq_relu = dq.relu;
        a.dq_bytes = (unsigned)((size_t)e.tH(qd) * e.tW(qd) * qd.Cs * 2);
        a.dst_bytes = 0;          // (the conv's own tensor is not stored)
    }
    dim3 grid(cdiv(a.Wout, TW) * cdiv(a.Hout, P->NW * (P->MT / 2)), P->nblocks_n);
    a.xq = PSEG_KNOB("PSEG_NO_XCD") ? -1 : (int)grid.x / 8;
    a.xr = (int)grid.x % 8;
    // persistent instance (PSEG_NO_PERSIST=1 disables): resident weights (NB == 1), single channel block, two
    // workgroups per CU walking 12 tiles each: the 38 KB weight set and the k-chunk table are staged once per
    // workgroup instead of once per tile.  Worth 1-3 % on the fused conv1+conv2 kernel (the DMA it saves was
    // mostly hidden by the co-resident workgroup); needs the per-trip opaque lane ids to keep two waves per SIMD.
    // wave-specialised persistent kernel (conv12_ws_kernel): one 512-thread workgroup per CU, two input tiles + the resident
    // weights in LDS.  PSEG_NO_WS=1 falls back to the every-wave-does-everything fused instance below.
    // streamed-weights persistent kernel with loader waves (conv_sp_kernel): conv5, conv6, conv7, deconv1 (+ deconv2), deconv3
    if (P->sp && !a.trace && !PSEG_KNOB("PSEG_NO_SP") && !PSEG_KNOB("PSEG_GENERIC") && (op.dq_fuse >= 0) == ((P->sp_fl & SP_DQ) != 0) &&
        (op.pool_dst >= 0) == ((P->sp_fl & SP_POOL) != 0)) {
        int dev = 0;
        const int cus_sp = device_cus(&dev);
        // Where it pays (same box, us per layer, conv_sp_kernel vs conv_mfma_kernel): the 80-channel layers always (2048x1536: conv7 20.5 vs
        // 27.0, deconv1 + deconv2 30.4 vs 39.3; 4096x3072: 48.7 vs 73.1, 75.3 vs 112.1 -- their five cout tiles leave conv_mfma_kernel two
        // workgroups per CU at best); the 40- / 60-channel layers while a launch is one round of tiles (1024x768: 14.7 / 19.7 / 26.7 vs
        // 19.4 / 25.5 / 37.0) -- with several tiles per CU three co-resident conv_mfma_kernel workgroups (three waves per SIMD, each
        // hiding the others' fragment reads) still beat one compute wave per SIMD by 3-10 %.  PSEG_SP_ALL=1: every eligible layer.
        const bool sp_pays = P->sp_NT == 5 || (int)grid.x * (e.batch_pages > 1 ? e.batch_pages : 1) <= cus_sp || PSEG_KNOB("PSEG_SP_ALL");
        SConv c{};
        c.src0 = a.src0; c.src1 = a.src1; c.nch0 = a.nch0; c.nch1 = a.nch1; c.bytes0 = a.bytes0; c.bytes1 = a.bytes1;
        c.Hin = a.Hin; c.Win = a.Win; c.Hout = a.Hout; c.Wout = a.Wout; c.pt = a.pt; c.pl = a.pl; c.relu = a.relu;
        c.nblk = P->sp_nblk; c.nc_full = P->sp_nc_full; c.nc_last = P->sp_nc_last; c.K = P->sp_K; c.S = P->sp_S; c.blk_steps = P->sp_blk_steps;
        c.tab = P->d_sp_tab; c.wpk = P->d_sp_wpk; c.bias = P->d_bias;
        c.row_pitch = P->sp_row_pitch; c.TBLK = P->sp_TBLK; c.RK = P->sp_RK;
        c.lds_ring_off = P->sp_ring_off; c.lds_patch_off = P->sp_patch_off; c.lds_tab_off = P->sp_tab_off; c.lds_flag_off = P->sp_flag_off;
        // tiles of 8 x 24 where they need fewer pixel columns per CU than tiles of 8 x 32: rounds of the launch x tile width (a
        // 2048x1536 page at 1/8 resolution: 192 wide tiles on 256 CUs = 32 columns each, or 256 narrow ones = 24)
        const int npg_ = e.batch_pages > 1 ? e.batch_pages : 1;
        const int tiles24 = cdiv(a.Wout, 24) * cdiv(a.Hout, 8);
        const bool narrow = P->sp24.ok && cdiv(tiles24 * npg_, cus_sp) * 24 < cdiv((int)grid.x * npg_, cus_sp) * 32;
        int tiles_pp = (int)grid.x, sp_lds = P->sp_lds;
        if (narrow) {
            const auto& N = P->sp24;
            c.tab = N.d_tab; c.row_pitch = N.row_pitch; c.TBLK = N.TBLK; c.RK = N.RK;
            c.lds_ring_off = N.ring_off; c.lds_patch_off = N.patch_off; c.lds_tab_off = N.tab_off; c.lds_flag_off = N.flag_off;
            tiles_pp = tiles24; sp_lds = N.lds;
        }
        c.dst = a.dst; c.dst_bytes = a.dst_bytes; c.nch_out = a.nch_out; c.pool_dst = a.pool_dst; c.pool_bytes = a.pool_bytes;
        c.dq_bias = a.dq_bias; c.dq_dst = a.dq_dst; c.dq_bytes = a.dq_bytes; c.dq_nch = a.dq_nch; c.dq_relu = a.dq_relu;
        const int npg = e.batch_pages > 1 ? e.batch_pages : 1;               // page slots of this launch: a tile index carries the page
        // ---- two compute teams (conv_sp2_kernel): the 40- / 60-channel layers once every CU gets a tile pair --------------------------
        // Tile width by rounds of pairs x width, as above: a 2048x1536 page at 1/4 resolution is 768 tiles of 8 x 32 (1.5 pairs per CU: two
        // rounds) or 1024 of 8 x 24 (exactly two pairs per CU).
        if ((!tracing_req(op) || PSEG_DIAG_KNOB("PSEG_SP2_TRACE")) && (P->sp2[0].ok || P->sp2[1].ok) && !sp2_off()) {
            const char* const sp2_sw = PSEG_KNOB("PSEG_SP2");   // plan switch (tests, A/B): 0 off, 1 every eligible launch, 24 / 32 likewise with that tile width
            int bestv = -1, best_cost = 1 << 30, best_tiles = 0;
            for (int v = 0; v < 2; ++v) {
                if (!P->sp2[v].ok) continue;
                const int twv = v == 0 ? 32 : 24;
                const int tl = cdiv(a.Wout, twv) * cdiv(a.Hout, 8);
                const int cost = cdiv((tl * npg + 1) / 2, cus_sp) * twv;
                if (cost < best_cost) { best_cost = cost; bestv = v; best_tiles = tl; }
            }
            if (sp2_sw && (atoi(sp2_sw) == 24 || atoi(sp2_sw) == 32)) {
                const int v = atoi(sp2_sw) == 24 ? 1 : 0;
                if (P->sp2[v].ok) { bestv = v; best_tiles = cdiv(a.Wout, v == 0 ? 32 : 24) * cdiv(a.Hout, 8); }
            }
            const int npairs = (best_tiles * npg + 1) / 2;
            // Where it runs by default: measured on the 2048x1536 page and in 32-page units against conv_mfma_kernel (tools/gpu_r05_sp2_ab.sh,
            // gpu_r05_sp2_pages.sh, us per page): deconv3 60.8-61.9 vs 60.6 / 47.7-48.4 vs 50.2-50.7; conv6 41.2-41.9 vs 41.8-42.3 / 33.1 vs
            // 30.2; conv5 32.1-32.5 vs 31.5-32.0 (before its table took the ring's room) -- a tie on the single page, a gain only for the
            // three-cout-tile layer in page units.  PSEG_SP2=1 / 24 / 32 (plan switch): every eligible layer.
            if (bestv >= 0 && ((npairs >= cus_sp && P->sp_NT == 3 && npg > 1) || sp2_sw)) {
                const auto& Q = P->sp2[bestv];
                SConv d = c;
                d.tab = Q.d_tab; d.TBLK = Q.TBLK; d.RK = Q.RK; d.row_pitch = sp_row_pitch(P->sp_sigma, bestv == 0 ? 32 : 24);
                d.lds_ring_off = Q.ring_off; d.lds_patch_off = Q.patch_off; d.lds_tab_off = Q.tab_off; d.lds_flag_off = Q.flag_off;
                d.ntiles = best_tiles * npg; d.tiles_per_page = best_tiles;
                d.xq = a.xq < 0 ? -1 : npairs / 8; d.xr = npairs % 8;
                d.err = e.d_sp_err;
                d.layer_id = (int)(&op - e.ops.data());
                d.dbg = PSEG_DIAG_KNOB("PSEG_SP_DBG") ? atoi(PSEG_DIAG_KNOB("PSEG_SP_DBG")) : 0;
                const dim3 g2((unsigned)std::min<int>(npairs, cus_sp));
                const int fl2 = P->sp_fl & SP_POOL;
                bool done2 = false;
                const bool trace2 = PSEG_DIAG_KNOB("PSEG_SP2_TRACE") && tracing_req(op);   // diagnostic build: stamps -> gpurun_out/sp2_trace_<layer>.bin
                if (trace2) {
                    PSEG_HIP(hipMalloc((void**)&d.trace, (size_t)g2.x * 16 * 8));
                    PSEG_HIP(hipMemset(d.trace, 0, (size_t)g2.x * 16 * 8));
                }
                if (PSEG_KNOB("PSEG_LOG_SP")) fprintf(stderr, "[pseg] conv_sp2 launch %s: tw %d tiles %d pairs %d grid %u RK %d lds %d\n", op.layer.c_str(), bestv ? 24 : 32, d.ntiles, npairs, g2.x, d.RK, Q.lds);
#define PSEG_SP2(NT_, SG_, FL_, TW_)                                                                                \
                if (!done2 && P->sp_NT == NT_ && P->sp_sigma == SG_ && fl2 == (FL_) && (bestv == 1) == (TW_ == 24)) {    \
                    static bool attr_set[64] = {false};                                                               \
                    if (!attr_set[dev & 63]) {                                                                        \
                        PSEG_HIP(hipFuncSetAttribute((const void*)conv_sp2_kernel<NT_, SG_, (FL_), TW_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
                        attr_set[dev & 63] = true;                                                                    \
                    }                                                                                                 \
                    conv_sp2_kernel<NT_, SG_, (FL_), TW_><<<g2, 768, Q.lds, st>>>(d);                                  \
                    PSEG_HIP(hipGetLastError());                                                                      \
                    done2 = true;                                                                                     \
                }
                PSEG_SP2(4, 5, 0, 24)           // conv5 (its 8 x 32 blocks leave no room for a ring)
                PSEG_SP2(4, 4, SP_POOL, 24) PSEG_SP2(4, 4, SP_POOL, 32)   // conv6
                PSEG_SP2(4, 4, 0, 24) PSEG_SP2(4, 4, 0, 32)
                PSEG_SP2(4, 5, SP_POOL, 24)
                PSEG_SP2(3, 4, 0, 24) PSEG_SP2(3, 4, 0, 32)               // deconv3
#undef PSEG_SP2
                if (done2 && trace2) {
                    PSEG_HIP(hipStreamSynchronize(st));
                    std::vector<unsigned long long> hbuf((size_t)g2.x * 16);
                    PSEG_HIP(hipMemcpy(hbuf.data(), d.trace, hbuf.size() * 8, hipMemcpyDeviceToHost));
                    (void)hipFree(d.trace);
                    const std::string fn = std::string("gpurun_out/sp2_trace_") + op.layer + ".bin";
                    if (FILE* f = fopen(fn.c_str(), "wb")) { fwrite(hbuf.data(), 8, hbuf.size(), f); fclose(f); }
                }
                if (done2) {
                    if (PSEG_KNOB("PSEG_SP_CHECK")) PSEG_TRY(engine_status(e, st));
                    return PSEG_OK;
                }
            }
        }
        c.ntiles = tiles_pp * npg; c.tiles_per_page = tiles_pp;
        c.xq = a.xq < 0 ? -1 : c.ntiles / 8; c.xr = c.ntiles % 8;
        c.err = e.d_sp_err;
        c.layer_id = (int)(&op - e.ops.data());
        c.dbg = PSEG_DIAG_KNOB("PSEG_SP_DBG") ? atoi(PSEG_DIAG_KNOB("PSEG_SP_DBG")) : 0;
        const dim3 gs((unsigned)std::min<int>(c.ntiles, cus_sp));
        const char* trl = PSEG_DIAG_KNOB("PSEG_SP_TRACE");     // diagnostic build only: the release library neither stamps nor writes files
        const bool tracing = sp_pays && trl && op.layer == trl;
        if (tracing) {
            PSEG_HIP(hipMalloc((void**)&c.trace, (size_t)gs.x * 16 * 8));
            PSEG_HIP(hipMemset(c.trace, 0, (size_t)gs.x * 16 * 8));
        }
        bool launched = !sp_pays;
#define PSEG_SP_TW(NT_, SG_, FL_, TW_)                                                                              \
        if (!launched && P->sp_NT == NT_ && P->sp_sigma == SG_ && P->sp_fl == (FL_) && narrow == (TW_ == 24)) {      \
            static bool attr_set[64] = {false};                                                                   \
            if (!attr_set[dev & 63]) {                                                                            \
                PSEG_HIP(hipFuncSetAttribute((const void*)conv_sp_kernel<NT_, SG_, (FL_), TW_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
                attr_set[dev & 63] = true;                                                                        \
            }                                                                                                     \
            conv_sp_kernel<NT_, SG_, (FL_), TW_><<<gs, 512, sp_lds, st>>>(c);                                      \
            PSEG_HIP(hipGetLastError());                                                                          \
            launched = true;                                                                                      \
        }
#define PSEG_SP(NT_, SG_, FL_) PSEG_SP_TW(NT_, SG_, FL_, 32)
        PSEG_SP(5, 4, SP_PATCH)                 // conv7
        PSEG_SP(5, 4, SP_DQ)                    // deconv1 + deconv2 (k2 s2) on its accumulators
        PSEG_SP_TW(5, 4, SP_PATCH, 24)          // ... and on tiles of 8 x 24 (one round of 256 tiles on a 2048x1536 page)
        PSEG_SP_TW(5, 4, SP_DQ, 24)
        PSEG_SP(4, 5, SP_PATCH)                 // conv5
        PSEG_SP(4, 4, SP_PATCH | SP_POOL)       // conv6
        PSEG_SP(4, 4, SP_PATCH)
        PSEG_SP(4, 5, SP_PATCH | SP_POOL)
        PSEG_SP(3, 4, 0)                        // deconv3 (fcn_skip: four channel blocks, no room for the store patch)
        PSEG_SP(3, 4, SP_PATCH)                 // deconv3 (fcn)
#undef PSEG_SP
#undef PSEG_SP_TW
        if (!sp_pays) launched = false;
        if (launched && tracing) {
            PSEG_HIP(hipStreamSynchronize(st));
            std::vector<unsigned long long> hbuf((size_t)gs.x * 16);
            PSEG_HIP(hipMemcpy(hbuf.data(), c.trace, hbuf.size() * 8, hipMemcpyDeviceToHost));
            (void)hipFree(c.trace);
            const std::string fn = std::string("gpurun_out/sp_trace_") + op.layer + ".bin";
            if (FILE* f = fopen(fn.c_str(), "wb")) { fwrite(hbuf.data(), 8, hbuf.size(), f); fclose(f); }
        }
        if (launched) {
            // tests (PSEG_SP_CHECK): the give-up record is looked at after EVERY launch; the product looks at it wherever the host
            // synchronises anyway (engine_status: pseg_predict, _batch, _chain, _exact_labels, pseg_engine_status)
            if (PSEG_KNOB("PSEG_SP_CHECK")) PSEG_TRY(engine_status(e, st));
            return PSEG_OK;
        }
    }
    // ping-pong persistent kernel (conv_pp_kernel) for the k5 mid layers whose weights fit LDS beside two tiles: conv3, conv4
    if (P->wg3 && P->KS == 5 && P->NT == 3 && P->MT == 4 && P->NW == 4 && P->nblk == 1 && P->nblocks_n == 1 && op.Cout <= 40 && !op.transposed &&
        op.src1 < 0 && !a.add && !a.in_relu && !a.up0 && op.fuse1 < 0 && op.tail_logits < 0 && op.skiplog < 0 && op.relu_dst < 0 &&
        P->pp && (a.sigma == 4 || a.sigma == 5) && !a.trace && !PSEG_KNOB("PSEG_NO_PP") && !PSEG_KNOB("PSEG_GENERIC")) {
        int dev = 0;
        const int cus_pp = device_cus(&dev);
        const int TB = round_up(P->THH * P->row_pitch, 16);
        const int lds = 2 * TB + P->ks_full * 2560 + 16 + round_up(P->ks_full * 16, 16);
        if (lds <= 160 * 1024 && (int)grid.x >= 2 * cus_pp) {
            MConv w = a;
            w.lds_w_off = TB;
            w.ntiles = (int)grid.x;
            if (PSEG_DIAG_KNOB("PSEG_PP_NODMA")) w.dbg |= 0x800;     // wrong results, timing only: diagnostic build only
            if (PSEG_DIAG_KNOB("PSEG_PP_NOEPI")) w.dbg |= 0x1000;
            const dim3 gp((unsigned)cus_pp);
#define PSEG_PP(SG_, POOL_)                                                                                       \
            if (a.sigma == SG_ && (a.pool_dst != nullptr) == POOL_) {                                             \
                static bool attr_set[64] = {false};                                                               \
                if (!attr_set[dev & 63]) {                                                                        \
                    PSEG_HIP(hipFuncSetAttribute((const void*)conv_pp_kernel<SG_, POOL_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
                    attr_set[dev & 63] = true;                                                                    \
                }                                                                                                 \
                conv_pp_kernel<SG_, POOL_><<<gp, 768, lds, st>>>(w);                                              \
                PSEG_HIP(hipGetLastError());                                                                      \
                return PSEG_OK;                                                                                   \
            }
            PSEG_PP(4, false) PSEG_PP(4, true) PSEG_PP(5, false) PSEG_PP(5, true)
#undef PSEG_PP
        }
    }
    if (op.fuse1 >= 0 && P->NB == 1 && P->nblk == 1 && P->nblocks_n == 1 && P->MT == 8 && P->NT == 2 && P->KS == 5 && a.sigma == 3 &&
        a.pool_dst && !a.add && !a.in_relu && !a.relu && !PSEG_KNOB("PSEG_NO_WS") && !PSEG_KNOB("PSEG_NO_PERSIST") && !PSEG_KNOB("PSEG_GENERIC") && !a.trace) {
        const bool ws_trace = PSEG_DIAG_KNOB("PSEG_WS_TRACE") != nullptr;   // developer aid: per-wave phase cycles -> gpurun_out/ws_trace.bin
        int dev = 0;
        const int cus_ws = device_cus(&dev);
        MConv w = a;
        const int TB = round_up(P->THH * P->row_pitch, 16);
        w.lds_w_off = TB;                                   // the kernel's tile stride
        w.ntiles = (int)grid.x;
        const int lds = WS_TILE0 + 2 * TB + P->ks_full * P->NT * 1024 + round_up(P->ks_full * 16, 16) + 4 * (2 * 9 * 96 + 16);
        const int form = PSEG_KNOB("PSEG_WS_FORM") ? atoi(PSEG_KNOB("PSEG_WS_FORM")) : 0;
        if (lds <= 160 * 1024 && P->GK >= P->ks_full && P->row_pitch == WS_ROWP && P->ks_full == (P->pairc2 ? WS_KSTEPS_PAIR : WS_KSTEPS)) {
            const unsigned gx = std::min<unsigned>(grid.x, (unsigned)cus_ws);
            const int sk = op.skiplog >= 0 ? 1 : 0;
            int variant = sk * 2 + (P->pairc2 ? 1 : 0);
            if (variant == 3 && (form == 1 || form == 2)) variant = 3 + form;       // the alternative tile loops exist for the default packing only
            const void* fn = variant == 5 ? (const void*)conv12_ws_kernel<true, true, 2> : variant == 4 ? (const void*)conv12_ws_kernel<true, true, 1>
                           : variant == 3 ? (const void*)conv12_ws_kernel<true, true, 0> : variant == 2 ? (const void*)conv12_ws_kernel<true, false, 0>
                           : variant == 1 ? (const void*)conv12_ws_kernel<false, true, 0> : (const void*)conv12_ws_kernel<false, false, 0>;
            static bool attr_ws[64][6] = {{false}};
            if (!attr_ws[dev & 63][variant]) {
                PSEG_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                attr_ws[dev & 63][variant] = true;
            }
            if (ws_trace) {
                PSEG_HIP(hipMalloc((void**)&w.trace, (size_t)gx * 8 * 4 * 8));
                PSEG_HIP(hipMemset(w.trace, 0, (size_t)gx * 8 * 4 * 8));
            }
            switch (variant) {
                case 5: conv12_ws_kernel<true, true, 2><<<dim3(gx), 512, lds, st>>>(w); break;
                case 4: conv12_ws_kernel<true, true, 1><<<dim3(gx), 512, lds, st>>>(w); break;
                case 3: conv12_ws_kernel<true, true, 0><<<dim3(gx), 512, lds, st>>>(w); break;
                case 2: conv12_ws_kernel<true, false, 0><<<dim3(gx), 512, lds, st>>>(w); break;
                case 1: conv12_ws_kernel<false, true, 0><<<dim3(gx), 512, lds, st>>>(w); break;
                default: conv12_ws_kernel<false, false, 0><<<dim3(gx), 512, lds, st>>>(w); break;
            }
            if (ws_trace) {
                PSEG_HIP(hipStreamSynchronize(st));
                std::vector<unsigned long long> hbuf((size_t)gx * 8 * 4);
                PSEG_HIP(hipMemcpy(hbuf.data(), w.trace, hbuf.size() * 8, hipMemcpyDeviceToHost));
                (void)hipFree(w.trace);
                if (FILE* f = fopen("gpurun_out/ws_trace.bin", "wb")) { fwrite(hbuf.data(), 8, hbuf.size(), f); fclose(f); }
            }
            return PSEG_OK;
        }
    }
    if (P->pairc2) return fail(PSEG_EUNSUPPORTED, "layer %s is packed for conv12_ws_kernel (paired half chunks), which cannot take this launch", op.layer.c_str());
    if (op.fuse1 >= 0 && P->NB == 1 && P->nblk == 1 && P->nblocks_n == 1 && !PSEG_KNOB("PSEG_NO_PERSIST") && !PSEG_KNOB("PSEG_GENERIC")) {
        const int cus = device_cus();
        a.ntiles = (int)grid.x;
        grid.x = std::min<unsigned>(grid.x, 2u * (unsigned)cus);
    }
    // ... and the k3 single-block layers whose instances exist (launch_generic_any2): many tiles, nine k-steps each
    if (op.fuse1 < 0 && P->NB == 1 && P->nblk == 1 && P->nblocks_n == 1 && P->KS == 3 && a.sigma == 4 && !a.deconv && !a.tail && !a.up0 && !a.up1 &&
        !a.pool_dst && !a.dst2 && !a.skip_logits && !a.tail_wa && !a.dq_w && e.batch_pages <= 1 &&
        P->MT == 2 && P->NT == 4 && op.stride == 2 && !a.add &&
        !PSEG_KNOB("PSEG_NO_PERSIST") && !PSEG_KNOB("PSEG_GENERIC")) {
        const int cus = device_cus();
        const unsigned slots = (P->wg3 ? 3u : 2u) * (unsigned)cus;
        if (grid.x > 2 * slots) { a.ntiles = (int)grid.x; grid.x = slots; }
    }
    if (e.batch_pages > 1) grid.z = (unsigned)e.batch_pages;   // (mfma_op_batchable layers only: the page slot is blockIdx.z)
    return launch_generic_any(a, *P, grid, st, op.layer.c_str());
}

// Layers whose kernel takes all page slots of a batch in ONE launch (run_bf16_pages): conv_sp_kernel (a tile index carries
// the page) and the plain conv_mfma_kernel instances (blockIdx.z = page slot; sources, output and fused pool only).  The
// others -- the fused first layers, the ping-pong kernel, the tails -- are launched once per page slot.
bool mfma_op_batchable(const Engine& e, const Op& op) {
    auto* P = (MfmaPlan*)op.plan;
    if (!P || P->kind != PLAN_GENERIC || op.type != OP_CONV || PSEG_KNOB("PSEG_NO_PAGE_LAUNCH")) return false;
    if (op.add >= 0 || op.relu_dst >= 0 || op.fuse1 >= 0 || op.tail_logits >= 0 || op.skiplog >= 0 || op.in_relu || op.up0 || op.up1 ||
        op.src0 == e.input_tensor || op.src1 == e.input_tensor)
        return false;
    if (P->pp) return false;
    // layer-major launches pay where one page leaves the chip partly filled or its launch is mostly ramp and drain: from 1/4 resolution
    // down (a 2048x1536 page: 768 tiles there).  The full- and half-resolution layers of the 3x3 graphs stay page-major: a page's
    // tensors are in the caches for its next layer.
    if (e.tensors[op.dst].s < 2) return false;
    if (op.dq_fuse >= 0) return P->sp && (P->sp_fl & SP_DQ) != 0;
    return true;
}

int mfma_launch_deconv2(Engine& e, Op& op, hipStream_t st) {
    auto* P = (MfmaPlan*)op.plan;
    if (!P) return fail(PSEG_EINVAL, "layer %s has no bf16 plan", op.layer.c_str());
    MConv a{};
    fill_common(e, op, *P, a);
    a.Hout = a.Hin;  // the GEMM pixel grid is the input grid
    a.Wout = a.Win;
    a.stride = 1;
    a.pt = a.pl = 0;
    a.deconv = 1;
    if (op.tail_logits >= 0) {
        const Op& lg = e.ops[op.tail_logits];
        if (P->tc_CP > 0 && !PSEG_KNOB("PSEG_GENERIC")) {
            // composed tail: one small GEMM per half-resolution pixel, no LDS
            const Tensor& s0 = e.tensors[op.src0];
            TailC t{};
            t.src0 = (const uint16_t*)s0.d;
            t.src1 = op.src1 >= 0 ? (const uint16_t*)e.tensors[op.src1].d : nullptr;
            t.skip = lg.src1 >= 0 ? (const uint16_t*)e.tensors[lg.src1].d : nullptr;
            t.nch0 = s0.Cs / 8;
            t.nch1 = op.src1 >= 0 ? e.tensors[op.src1].Cs / 8 : 0;
            t.nch_skip = lg.src1 >= 0 ? e.tensors[lg.src1].Cs / 8 : 0;
            t.Hh = e.tH(s0); t.Wh = e.tW(s0);
            t.H0 = e.H; t.W0 = e.W; t.C = lg.Cout;
            t.wA1 = P->d_tc_wA1; t.wA2 = P->d_tc_wA2; t.beta = P->d_tc_beta;
            if (lg.src1 >= 0) {
                const int sp = producer_of(e, lg.src1);
                if (sp >= 0 && e.ops[sp].skiplog >= 0) {
                    auto* PS = (MfmaPlan*)e.ops[sp].plan;
                    if (!PS || !PS->d_skiplog || PS->skip_CP != P->tc_CP) return fail(PSEG_EINVAL, "skip-logits buffer missing for the composed tail");
                    t.S = (const float*)((const char*)PS->d_skiplog + (size_t)e.page * ((size_t)e.Hp * e.Wp * PS->skip_CP * 4));   // this page slot's plane
                    t.skip = nullptr;
                }
            }
            t.out_logits = e.cur_logits; t.out_probs = e.cur_probs; t.out_labels = e.cur_labels; t.out_labels_u8 = e.cur_labels_u8;
            t.out_margin = e.cur_margin;
            e.margin_done = e.cur_margin != nullptr;
            if (t.Wh % 16) return fail(PSEG_EINVAL, "composed tail needs a canvas width multiple of 32");
            if (P->tail2) {
                const Op& dq = e.ops[producer_of(e, op.src0)];
                auto* PQ = (MfmaPlan*)dq.plan;
                if (!PQ || !PQ->d_q_w || (!t.S && lg.src1 >= 0) || (P->tc_CP != 4 && P->tc_CP != 8)) return fail(PSEG_EINVAL, "fused inner deconv: plan data missing");
                Tail2 u{};
                const Tensor& q0 = e.tensors[dq.src0];
                u.q0 = (const uint16_t*)q0.d; u.nq0 = q0.Cs / 8;
                u.q1 = dq.src1 >= 0 ? (const uint16_t*)e.tensors[dq.src1].d : nullptr;
                u.nq1 = dq.src1 >= 0 ? e.tensors[dq.src1].Cs / 8 : 0;
                u.c3 = t.src1; u.nc3 = t.nch1;
                u.S = t.S; u.Hh = t.Hh; u.Wh = t.Wh; u.H0 = t.H0; u.W0 = t.W0; u.C = t.C;
                u.wQ = PQ->d_q_w; u.biasQ = PQ->d_q_bias; u.wD = P->d_t2_wD; u.wC = P->d_t2_wC; u.beta = P->d_tc_beta;
                u.out_logits = t.out_logits; u.out_probs = t.out_probs; u.out_labels = t.out_labels; u.out_labels_u8 = t.out_labels_u8;
                u.out_margin = t.out_margin;
                const int nwg = cdiv(t.Hh, 2) * cdiv((t.Wh + 31) / 32, T2_ITER);      // one workgroup = 2 rows x 2 column parities
                if (P->tc_CP == 4) tail_fused2_kernel<4><<<nwg, 256, 0, st>>>(u);
                else tail_fused2_kernel<8><<<nwg, 256, 0, st>>>(u);
                PSEG_HIP(hipGetLastError());
                return PSEG_OK;
            }
            const int waves = t.Hh * (t.Wh / 16);
            const dim3 grid(cdiv(waves, 4));
            const int key = P->tc_CP * 100 + P->tc_nks0 * 10 + P->tc_nkss;
            switch (key) {
                case 430: tail_composed_kernel<4, 3, 0><<<grid, 256, 0, st>>>(t); break;   // skip logits from the producer's buffer
                case 830: tail_composed_kernel<8, 3, 0><<<grid, 256, 0, st>>>(t); break;
                case 431: tail_composed_kernel<4, 3, 1><<<grid, 256, 0, st>>>(t); break;
                case 831: tail_composed_kernel<8, 3, 1><<<grid, 256, 0, st>>>(t); break;
                case 1631: tail_composed_kernel<16, 3, 1><<<grid, 256, 0, st>>>(t); break;
                case 410: tail_composed_kernel<4, 1, 0><<<grid, 256, 0, st>>>(t); break;
                case 810: tail_composed_kernel<8, 1, 0><<<grid, 256, 0, st>>>(t); break;
                case 1610: tail_composed_kernel<16, 1, 0><<<grid, 256, 0, st>>>(t); break;
                default: return fail(PSEG_EUNSUPPORTED, "no composed-tail instance %d", key);
            }
            PSEG_HIP(hipGetLastError());
            return PSEG_OK;
        }
        a.tail = 1;
        a.tail_C = lg.Cout;
        a.H0 = e.H;
        a.W0 = e.W;
        a.skip = lg.src1 >= 0 ? (const uint16_t*)e.tensors[lg.src1].d : nullptr;
        a.nch_skip = lg.src1 >= 0 ? e.tensors[lg.src1].Cs / 8 : 0;
        a.tail_wa = P->d_tail_wa;
        a.tail_wb = P->d_tail_wb;
        a.tail_bias = P->d_tail_bias;
        a.out_logits = e.cur_logits;
        a.out_probs = e.cur_probs;
        a.out_labels = e.cur_labels;
        a.out_labels_u8 = e.cur_labels_u8;
        // the specialised tail instances walk both N blocks in one workgroup (NBL = 2 in the kernel)
        if (P->nblk == 1 && P->nblocks_n == 2 && !PSEG_KNOB("PSEG_GENERIC") && (a.sigma == 10 || a.sigma == 6)) a.nb_loop = 2;
    }
    if (op.tail_logits < 0 && P->nblk == 1 && P->nblocks_n == 2 && !PSEG_KNOB("PSEG_GENERIC") && a.sigma == 14 && P->NT == 4) a.nb_loop = 2;   // deconv4 (fcn_skip)
    dim3 grid(cdiv(a.Wout, TW) * cdiv(a.Hout, 2 * P->MT), P->nblocks_n / a.nb_loop);
    a.xq = PSEG_KNOB("PSEG_NO_XCD") ? -1 : (int)grid.x / 8;
    a.xr = (int)grid.x % 8;
    return launch_generic_any(a, *P, grid, st, op.layer.c_str());
}

int mfma_launch_pool(Engine& e, Op& op, hipStream_t st) {
    const Tensor& s = e.tensors[op.src0];
    const size_t n = (size_t)(e.tH(s) / 2) * (e.tW(s) / 2) * (s.Cs / 8);
    pool_bf16_kernel<<<(int)std::min<size_t>((n + 255) / 256, 8192), 256, 0, st>>>(
        (const uint16_t*)s.d, e.tH(s), e.tW(s), s.Cs / 8, (uint16_t*)e.tensors[op.dst].d);
    return PSEG_OK;
}

int mfma_launch_logits(Engine& e, Op& op, float* d_logits, float* d_probs, int64_t* d_labels,
                       uint8_t* d_labels_u8, hipStream_t st) {
    auto* P = (MfmaPlan*)op.plan;
    if (!P) return fail(PSEG_EINVAL, "logits layer has no bf16 plan");
    const Tensor& s0 = e.tensors[op.src0];
    const Tensor* s1 = op.src1 >= 0 ? &e.tensors[op.src1] : nullptr;
    const int grid = cdiv(e.H * e.W, 256);
    const uint16_t* p0 = (const uint16_t*)s0.d;
    const uint16_t* p1 = s1 ? (const uint16_t*)s1->d : nullptr;
    const int n0 = s0.Cs / 8, n1 = s1 ? s1->Cs / 8 : 0;
    if (P->cmax <= 16) {
        const int waves = e.H * cdiv(e.W, 16);
        logits_mfma_kernel<<<cdiv(waves, 4), 256, 0, st>>>(p0, n0, p1, n1, e.Wp, e.H, e.W, op.Cout, P->d_wpk, P->d_bias,
                                                          d_logits, d_probs, d_labels, d_labels_u8);
        PSEG_HIP(hipGetLastError());
        return PSEG_OK;
    }
#define LG(CM) logits_bf16_kernel<CM><<<grid, 256, 0, st>>>(p0, n0, p1, n1, e.Wp, e.H, e.W, P->d_wf, P->d_bias, \
                                                              op.Cout, d_logits, d_probs, d_labels, d_labels_u8)
    if (P->cmax == 4) LG(4);
    else if (P->cmax == 8) LG(8);
    else if (P->cmax == 16) LG(16);
    else if (P->cmax == 32) LG(32);
    else LG(64);
#undef LG
    return PSEG_OK;
}

bool mfma_tail_emits_margin(const Engine& e) {
    if (PSEG_KNOB("PSEG_GENERIC")) return false;
    for (auto& op : e.ops)
        if (op.type == OP_DECONV2 && op.tail_logits >= 0 && !op.fused_away) {
            auto* P = (MfmaPlan*)op.plan;
            return P && P->tc_CP > 0;
        }
    return false;
}

// The fused first-layer kernels read the uint8 page directly (e.cur_img); graphs whose first
// conv is not on that path get a bf16 canvas of x/255.
int mfma_preprocess(Engine& e, const uint8_t* d_img, hipStream_t st) {
    e.cur_img = d_img;
    bool all_special = true;
    for (auto& op : e.ops)
        if (op.src0 == e.input_tensor || op.src1 == e.input_tensor) {
            auto* P = (MfmaPlan*)op.plan;
            if (!P || P->kind != PLAN_CONV1) all_special = false;
        }
    if (all_special) return PSEG_OK;
    Tensor& in = e.tensors[e.input_tensor];
    const size_t n = (size_t)e.Hp * e.Wp;
    preprocess_bf16_kernel<<<(int)std::min<size_t>((n + 255) / 256, 8192), 256, 0, st>>>(
        d_img, e.H, e.W, in.C, e.d_lut, (uint16_t*)in.d, e.Hp, e.Wp, in.Cs);
    return PSEG_OK;
}

}  // namespace pseg
