// pseg_upsplit.hip -- UpSampling2D(2) -> Conv2D(k2, 'same') (unet, lib/model.py:174-175,180-181,186-187) in split form.
//
// out(y, x) = sum_{a,b} W[a][b] . src((y+a) >> 1, (x+b) >> 1): every tap of every output pixel is a product
// P_ab(Y, X) = W[a][b] . src(Y, X) of a SOURCE pixel, and each P_ab(Y, X) is used by (up to) four output pixels.
// Computing the four products once per source pixel is a plain GEMM  D[M][4 Cout] = src[M][Cin] . Wg^T  with a
// quarter of the direct form's MACs (Cin Cout per output pixel instead of 4 Cin Cout); the output is then a
// 4-term gather-sum over D (+ bias, ReLU), an HBM-bound pass.  The GEMM is upsplit_gemm_kernel below (hand-written MFMA
// kernel, bf16 operands, f32 accumulation, both operands K-contiguous); the partial products are stored as
// bf16, one more rounding than the direct kernel (same class as the per-layer activation rounding; the bf16
// parity tests cover it).  That two-pass form (plan switch PSEG_UPSPLIT_TWO_PASS) served the deep layers (Cin >= 256); the default
// now are the one-pass kernels below -- the products of a patch of source pixels stay in LDS (upsplit_fused_kernel) or in registers
// (upsplit_reg_kernel, short GEMMs) and the same workgroup writes the outputs -- for every upsample -> k2 layer of unet: 2048x1536,
// same box, two-pass (direct kernel for the last) -> one pass: 1024->512 80 -> 80 us, 512->256 121 -> 94, 256->128 191 -> 141,
// 128->64 258 -> 200; unet page 4.72 -> 4.58 ms.
#include <algorithm>
#include <cstring>

#include "pseg_common.h"

namespace pseg {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

struct UpSplit {
    int Cs0 = 0, CoS = 0;
    uint16_t* d_w = nullptr;     // [4 CoS][Cs0] bf16, row n = (a*2+b) * CoS + co
    float* d_bias = nullptr;     // [CoS]
    uint16_t* d_D = nullptr;     // [M][4 CoS] bf16 partial products
    size_t D_bytes = 0;
};

void upsplit_free(UpSplit* u) {
    if (!u) return;
    (void)hipFree(u->d_w); (void)hipFree(u->d_bias); (void)hipFree(u->d_D);
    delete u;
}

// ---------------------------------------------------------------------------------------------------------------------
// D[M][N] = A[M][K] . B[N][K]^T, bf16 operands (both K-contiguous), float32 accumulation on v_mfma_f32_16x16x32_bf16,
// bf16 result.  A = the source pixels (M = Hs Ws, K = Cin as stored), B = the regrouped kernel (N = 4 Cout).
// One 256-thread workgroup per 128 x 128 tile of D, K in steps of 64 through a two-stage LDS ring filled by LDS-DMA
// (buffer_load ... lds: rows past M / N and chunks past K use an out-of-range offset, for which the hardware writes
// zeros).  An operand tile is [128 rows][8 chunks of 16 B]; the DMA lets every lane pick its SOURCE chunk freely, so
// the tile is stored XOR-swizzled -- slot (row, c ^ (row & 7)) holds chunk c -- which makes the MFMA fragment reads
// (lane (p16, g) reads chunk 4 ks + g of row p16: one ds_read_b128) conflict-free without padding.  A wave owns a
// 64 x 64 sub-tile: 4 x 4 accumulator tiles, per 32-deep k-step four pixel fragments and four kernel fragments for
// sixteen MFMAs, the reads of the next k-step issued between the MFMAs of the current one.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int GM = 128, GN = 128, GK = 64;

__device__ __forceinline__ void gemm_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__global__ __launch_bounds__(256, 2) void upsplit_gemm_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B,
                                                              uint16_t* __restrict__ D, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];      // [2 stages][A tile 16 KiB | B tile 16 KiB]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p16 = lane & 15, g = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * GM, n0 = blockIdx.y * GN;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (unsigned)((size_t)M * K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (unsigned)((size_t)N * K * 2), 0x00020000);
    constexpr unsigned OOB = 0xfffffff0u;
    // DMA pieces of this wave: piece jj = wave * 4 + j fills slots jj * 64 .. + 63 of a tile; slot s = (row s >> 3, position s & 7)
    unsigned offA[4], offB[4], chk[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int sl = (wave * 4 + j) * 64 + lane, row = sl >> 3, c = (sl & 7) ^ (row & 7);
        chk[j] = (unsigned)c;
        offA[j] = m0 + row < M ? (unsigned)(((size_t)(m0 + row) * K + c * 8) * 2) : OOB;
        offB[j] = n0 + row < N ? (unsigned)(((size_t)(n0 + row) * K + c * 8) * 2) : OOB;
    }
    auto stage = [&](int k0, int buf) {
        char* base = smem + buf * 32768 + wave * 4096;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool kin = k0 + (int)chk[j] * 8 < K;                      // a chunk past K reads zeros
            const unsigned oa = (offA[j] == OOB || !kin) ? OOB : offA[j] + (unsigned)k0 * 2u;
            const unsigned ob = (offB[j] == OOB || !kin) ? OOB : offB[j] + (unsigned)k0 * 2u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(base + j * 1024), 16, oa, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(base + 16384 + j * 1024), 16, ob, 0, 0, 0);
        }
    };
    const int nk = (K + GK - 1) / GK;
    stage(0, 0);
    if (nk > 1) stage(GK, 1);
    f32x4 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    // fragment addresses inside a stage: row * 128 + ((4 ks + g) ^ (p16 & 7)) * 16
    const int sw0 = ((g) ^ (p16 & 7)) * 16, sw1 = ((4 + g) ^ (p16 & 7)) * 16;
    const int rowA = (wm * 64 + p16) * 128, rowB = 16384 + (wn * 64 + p16) * 128;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        gemm_lds_barrier();                                                   // stage kt has landed for every wave
        const char* sb = smem + (kt & 1) * 32768;
        bf16x8 xf[2][4], wf[2][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            xf[0][i] = *(const bf16x8*)(sb + rowA + i * 2048 + sw0);
            wf[0][i] = *(const bf16x8*)(sb + rowB + i * 2048 + sw0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            xf[1][i] = *(const bf16x8*)(sb + rowA + i * 2048 + sw1);
            wf[1][i] = *(const bf16x8*)(sb + rowB + i * 2048 + sw1);
        }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][ni], xf[0][mi], acc[mi][ni], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 8; ++q) {                                          // the second k-step's reads ride behind the first's MFMAs
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][ni], xf[1][mi], acc[mi][ni], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 2 < nk) {
            gemm_lds_barrier();                                               // every wave is done reading this buffer
            stage((kt + 2) * GK, kt & 1);
        }
    }
    // D layout of an accumulator tile: lane (p16, g) holds pixel p16 x columns 4g .. 4g+3 (8 bytes).  The wave's 64 x 64 bf16
    // block goes through its quarter of the (now idle) LDS ring so that every store instruction writes whole 128-byte rows:
    // 8 lanes x 16 B per row, 8 rows per instruction.
    gemm_lds_barrier();                                                       // every wave is done with the last stage
    char* tr = smem + wave * 8192;                                            // [64 rows][128 B]
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const f32x4 v = acc[mi][ni];
            uint2 pk;
            pk.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{v[0], v[1]}, b2));
            pk.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{v[2], v[3]}, b2));
            // 16-byte slot (ni*2 + g/2) of row r, XOR-swizzled by the row so that the 16 rows of a tile spread over the banks
            const int r = mi * 16 + p16, slot = (ni * 2 + (g >> 1)) ^ (r & 7);
            *(uint2*)(tr + r * 128 + slot * 16 + (g & 1) * 8) = pk;
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                        // the wave reads back only what it wrote itself
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int r = it * 8 + (lane >> 3), c = lane & 7;
        const uint4 v = *(const uint4*)(tr + r * 128 + ((c ^ (r & 7)) * 16));
        const int m = m0 + wm * 64 + r, n = n0 + wn * 64 + c * 8;
        if (m < M && n < N) *(uint4*)(D + (size_t)m * N + n) = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same products WITHOUT the trip through HBM: one workgroup takes an 8 x 16 patch of source pixels (the GEMM's 128 rows)
// and 32 output channels (128 columns: 4 taps x 32), keeps the 128 x 128 float32 products in the LDS the operand ring no longer
// needs and writes the 14 x 30 output pixels whose four terms all lie inside the patch -- patches step by 7 x 15 source pixels,
// so 22 % of the products are computed twice, which is cheap (the GEMM has a quarter of the direct form's MACs) next to what
// the split form moved: D written and gathered again, 0.8-1.4 GB per layer at 1/2 and full resolution against 0.3-0.6 GB here.
// The products stay float32 until the sum (the split form rounded them to bf16): one rounding per output, as the direct kernel.
// Source pixels outside the image are the descriptor's zeros = the 'same' padding of the k2 kernel on the upsampled map.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int UF_TY = 8, UF_TX = 16;                      // source patch (rows of the GEMM: r = ty * 16 + tx)
__global__ __launch_bounds__(256, 2) void upsplit_fused_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B,
                                                               const float* __restrict__ bias, uint16_t* __restrict__ dst,
                                                               int Hs, int Ws, int K, int CoS, int relu) {
    extern __shared__ __attribute__((aligned(16))) char smem[];      // [2 stages][A tile 16 KiB | B tile 16 KiB], then D [128][128] f32
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p16 = lane & 15, g = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntx = (Ws + UF_TX - 2) / (UF_TX - 1);
    const int nb = blockIdx.x, tile = blockIdx.y;          // cout block fastest: the blocks of a patch run together and share its pixels in L2
    const int Y0 = (tile / ntx) * (UF_TY - 1), X0 = (tile % ntx) * (UF_TX - 1);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (unsigned)((size_t)Hs * Ws * K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (unsigned)((size_t)4 * CoS * K * 2), 0x00020000);
    constexpr unsigned OOB = 0xfffffff0u;
    unsigned offA[4], offB[4], chk[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int sl = (wave * 4 + j) * 64 + lane, row = sl >> 3, c = (sl & 7) ^ (row & 7);
        chk[j] = (unsigned)c;
        const int y = Y0 + (row >> 4), x = X0 + (row & 15);
        offA[j] = (y < Hs && x < Ws) ? (unsigned)((((size_t)y * Ws + x) * K + c * 8) * 2) : OOB;
        const int ab = row >> 5, co = nb * 32 + (row & 31);                     // column n = tap * 32 + output channel of the block
        offB[j] = co < CoS ? (unsigned)((((size_t)ab * CoS + co) * K + c * 8) * 2) : OOB;
    }
    auto stage = [&](int k0, int buf) {
        char* base = smem + buf * 32768 + wave * 4096;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool kin = k0 + (int)chk[j] * 8 < K;
            const unsigned oa = (offA[j] == OOB || !kin) ? OOB : offA[j] + (unsigned)k0 * 2u;
            const unsigned ob = (offB[j] == OOB || !kin) ? OOB : offB[j] + (unsigned)k0 * 2u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(base + j * 1024), 16, oa, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(base + 16384 + j * 1024), 16, ob, 0, 0, 0);
        }
    };
    const int nk = (K + GK - 1) / GK;
    stage(0, 0);
    if (nk > 1) stage(GK, 1);
    f32x4 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int sw0 = ((g) ^ (p16 & 7)) * 16, sw1 = ((4 + g) ^ (p16 & 7)) * 16;
    const int rowA = (wm * 64 + p16) * 128, rowB = 16384 + (wn * 64 + p16) * 128;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        gemm_lds_barrier();
        const char* sb = smem + (kt & 1) * 32768;
        bf16x8 xf[2][4], wf[2][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            xf[0][i] = *(const bf16x8*)(sb + rowA + i * 2048 + sw0);
            wf[0][i] = *(const bf16x8*)(sb + rowB + i * 2048 + sw0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            xf[1][i] = *(const bf16x8*)(sb + rowA + i * 2048 + sw1);
            wf[1][i] = *(const bf16x8*)(sb + rowB + i * 2048 + sw1);
        }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][ni], xf[0][mi], acc[mi][ni], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][ni], xf[1][mi], acc[mi][ni], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 2 < nk) {
            gemm_lds_barrier();
            stage((kt + 2) * GK, kt & 1);
        }
    }
    // the products: row r (source pixel) x 32 slots of four floats, slot s of row r at position s ^ (r & 31) -- an accumulator
    // tile's 16 pixels write 16 different positions, the readers below (one row, consecutive slots per lane) stay conflict-free
    gemm_lds_barrier();                                                       // every wave is done with the last stage
    float* const Dl = (float*)smem;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int r = wm * 64 + mi * 16 + p16, sl = wn * 16 + ni * 4 + g;
            *(f32x4*)(Dl + r * 128 + ((sl ^ (r & 31)) << 2)) = acc[mi][ni];
        }
    gemm_lds_barrier();
    // outputs (2 (Y0 + ty) + i, 2 (X0 + tx) + j), ty < 7, tx < 15: bias + the four taps in order 00, 01, 10, 11.  A thread takes the
    // 2 x 2 outputs of ONE source pixel for eight channels: nine products (18 reads of four floats) instead of sixteen
    const int Ho = 2 * Hs, Wo = 2 * Ws;
    for (int it = tid; it < (UF_TY - 1) * (UF_TX - 1) * 4; it += 256) {
        const int c8 = it & 3, px = it >> 2, ty = px / (UF_TX - 1), tx = px - ty * (UF_TX - 1);
        const int co = nb * 32 + c8 * 8;
        if (2 * (Y0 + ty) >= Ho || 2 * (X0 + tx) >= Wo || co >= CoS) continue;
        auto ld = [&](int ab, int dy, int dx, float* v) {
            const int r = (ty + dy) * 16 + tx + dx, s0 = ab * 8 + c8 * 2;
            const f32x4 v0 = *(const f32x4*)(Dl + r * 128 + ((s0 ^ (r & 31)) << 2));
            const f32x4 v1 = *(const f32x4*)(Dl + r * 128 + (((s0 + 1) ^ (r & 31)) << 2));
            v[0] = v0[0]; v[1] = v0[1]; v[2] = v0[2]; v[3] = v0[3]; v[4] = v1[0]; v[5] = v1[1]; v[6] = v1[2]; v[7] = v1[3];
        };
        float p00[8], p01[2][8], p10[2][8], p11[2][2][8], bb[8];
        ld(0, 0, 0, p00);
        ld(1, 0, 0, p01[0]); ld(1, 0, 1, p01[1]);
        ld(2, 0, 0, p10[0]); ld(2, 1, 0, p10[1]);
        ld(3, 0, 0, p11[0][0]); ld(3, 0, 1, p11[0][1]); ld(3, 1, 0, p11[1][0]); ld(3, 1, 1, p11[1][1]);
        const float4 b0 = *(const float4*)(bias + co), b1 = *(const float4*)(bias + co + 4);
        bb[0] = b0.x; bb[1] = b0.y; bb[2] = b0.z; bb[3] = b0.w; bb[4] = b1.x; bb[5] = b1.y; bb[6] = b1.z; bb[7] = b1.w;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int y = 2 * (Y0 + ty) + i, x = 2 * (X0 + tx) + j;
                if (y >= Ho || x >= Wo) continue;
                uint32_t o[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float a0 = (((bb[2 * q] + p00[2 * q]) + p01[j][2 * q]) + p10[i][2 * q]) + p11[i][j][2 * q];
                    float a1 = (((bb[2 * q + 1] + p00[2 * q + 1]) + p01[j][2 * q + 1]) + p10[i][2 * q + 1]) + p11[i][j][2 * q + 1];
                    if (relu) { a0 = a0 > 0.f ? a0 : 0.f; a1 = a1 > 0.f ? a1 : 0.f; }
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
                    o[q] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{a0, a1}, b2));
                }
                *(uint4*)(dst + ((size_t)y * Wo + x) * CoS + co) = make_uint4(o[0], o[1], o[2], o[3]);
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// ... and WITHOUT the trip through LDS.  The GEMM's columns are ordered so that the four accumulator values of a lane are the
// four TAPS of one output channel -- column n = 16 nt + 4 g + t holds (tap t, channel 8 g + nt): lane (pixel p16, g) ends up
// with the products of channels 8 g .. 8 g + 7 (its eight column tiles) of its pixel, 16 contiguous bytes of every output.
// A wave owns three source rows of 16 pixels (rows 2 w, 2 w + 1 and the halo row 2 w + 2 of a 9 x 16 patch: the even rows
// between waves are multiplied twice), so the terms of an output that belong to the row below are the lane's own registers
// (next row tile) and the terms of the pixel to the right come from the neighbouring lane: no product leaves the registers.
// A workgroup writes the 16 x 30 outputs of 8 x 15 source pixels for 32 channels; LDS holds the operand ring only.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int UR_ROWS = 9, UR_APIECES = UR_ROWS * 16 * 8 / 64, UR_PIECES = UR_APIECES + 16;   // 64-slot DMA pieces of a stage: 18 of A, 16 of B
constexpr int UR_STAGE = UR_PIECES * 1024;                                                   // 34 KiB
__global__ __launch_bounds__(256, 2) void upsplit_reg_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B,
                                                             const float* __restrict__ bias, uint16_t* __restrict__ dst,
                                                             int Hs, int Ws, int K, int CoS, int relu) {
    extern __shared__ __attribute__((aligned(16))) char smem[];      // [2 stages][A: 144 pixel rows x 128 B | B: 128 column rows x 128 B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p16 = lane & 15, g = lane >> 4;
    const int ntx = (Ws + 14) / 15;
    const int nb = blockIdx.x, tile = blockIdx.y;
    const int Y0 = (tile / ntx) * 8, X0 = (tile % ntx) * 15;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (unsigned)((size_t)Hs * Ws * K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (unsigned)((size_t)4 * CoS * K * 2), 0x00020000);
    constexpr unsigned OOB = 0xfffffff0u;
    // pieces wave, wave + 4, ... of a stage (nine for waves 0 and 1, eight for the others)
    unsigned off[9], chk[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const int piece = wave + 4 * j;
        off[j] = OOB; chk[j] = 0;
        if (piece < UR_PIECES) {
            const bool isA = piece < UR_APIECES;
            const int sl = (isA ? piece : piece - UR_APIECES) * 64 + lane, row = sl >> 3, c = (sl & 7) ^ (row & 7);
            chk[j] = (unsigned)c;
            if (isA) {
                const int y = Y0 + (row >> 4), x = X0 + (row & 15);
                if (y < Hs && x < Ws) off[j] = (unsigned)((((size_t)y * Ws + x) * K + c * 8) * 2);
            } else {
                const int t = row & 3, co = nb * 32 + ((row >> 2) & 3) * 8 + (row >> 4);   // column row = 16 nt + 4 g + t
                if (co < CoS) off[j] = (unsigned)((((size_t)t * CoS + co) * K + c * 8) * 2);
            }
        }
    }
    auto stage = [&](int k0, int buf) {
        char* base = smem + buf * UR_STAGE;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const int piece = wave + 4 * j;
            if (piece < UR_PIECES) {                                         // (uniform)
                const bool kin = k0 + (int)chk[j] * 8 < K;
                const unsigned o = (off[j] == OOB || !kin) ? OOB : off[j] + (unsigned)k0 * 2u;
                if (piece < UR_APIECES) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(base + piece * 1024), 16, o, 0, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(base + piece * 1024), 16, o, 0, 0, 0);
            }
        }
    };
    const int nk = (K + GK - 1) / GK;
    stage(0, 0);
    if (nk > 1) stage(GK, 1);
    f32x4 acc[3][8];
#pragma unroll
    for (int mi = 0; mi < 3; ++mi)
#pragma unroll
        for (int ni = 0; ni < 8; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int sw0 = ((g) ^ (p16 & 7)) * 16, sw1 = ((4 + g) ^ (p16 & 7)) * 16;
    const int rowA = ((2 * wave) * 16 + p16) * 128, rowB = UR_APIECES * 1024 + p16 * 128;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) { if (wave < 2) asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        gemm_lds_barrier();
        const char* sb = smem + (kt & 1) * UR_STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int sw = ks ? sw1 : sw0;
            bf16x8 xf[3], wf[8];
#pragma unroll
            for (int i = 0; i < 3; ++i) xf[i] = *(const bf16x8*)(sb + rowA + i * 2048 + sw);
#pragma unroll
            for (int i = 0; i < 8; ++i) wf[i] = *(const bf16x8*)(sb + rowB + i * 2048 + sw);
#pragma unroll
            for (int ni = 0; ni < 8; ++ni)
#pragma unroll
                for (int mi = 0; mi < 3; ++mi)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], xf[mi], acc[mi][ni], 0, 0, 0);
        }
        if (kt + 2 < nk) {
            gemm_lds_barrier();
            stage((kt + 2) * GK, kt & 1);
        }
    }
    // acc[mi][ni][t]: source pixel (row 2 wave + mi, column p16), channel 8 g + ni, tap t = 2 a + b
    const int Wo = 2 * Ws;
    const int cob = nb * 32 + g * 8;
    float bb[8];
    {
        const bool okc = cob < CoS;
        const float4 b0 = okc ? *(const float4*)(bias + cob) : make_float4(0.f, 0.f, 0.f, 0.f), b1 = okc ? *(const float4*)(bias + cob + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        bb[0] = b0.x; bb[1] = b0.y; bb[2] = b0.z; bb[3] = b0.w; bb[4] = b1.x; bb[5] = b1.y; bb[6] = b1.z; bb[7] = b1.w;
    }
    // the right neighbour's b = 1 terms: tap 01 of rows 0, 1, tap 11 of rows 0 .. 2
    float r01[2][8], r11[3][8];
#pragma unroll
    for (int ni = 0; ni < 8; ++ni) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) r01[mi][ni] = __shfl_down(acc[mi][ni][1], 1, 16);
#pragma unroll
        for (int mi = 0; mi < 3; ++mi) r11[mi][ni] = __shfl_down(acc[mi][ni][3], 1, 16);
    }
    if (p16 < 15 && cob < CoS && X0 + p16 < Ws) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int sy = Y0 + 2 * wave + mi;
            if (sy >= Hs) continue;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int y = 2 * sy + i, x = 2 * (X0 + p16) + j;
                    uint32_t o[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float v2[2];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int ni = 2 * q + h;
                            float v = ((bb[ni] + acc[mi][ni][0]) + (j ? r01[mi][ni] : acc[mi][ni][1])) + acc[mi + i][ni][2];
                            v += j ? r11[mi + i][ni] : acc[mi + i][ni][3];
                            v2[h] = relu ? (v > 0.f ? v : 0.f) : v;
                        }
                        typedef float f2 __attribute__((ext_vector_type(2)));
                        typedef __bf16 b2 __attribute__((ext_vector_type(2)));
                        o[q] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{v2[0], v2[1]}, b2));
                    }
                    *(uint4*)(dst + ((size_t)y * Wo + x) * CoS + cob) = make_uint4(o[0], o[1], o[2], o[3]);
                }
        }
    }
}

static inline uint16_t h_f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

// w: correlation-form f32 kernel [2][2][Cin][Cout]; Cs0 / CoS: channel counts as stored (multiples of 8)
int upsplit_create(UpSplit** out, const std::vector<float>& w, const std::vector<float>& bias, int Cin, int Cs0,
                   int Cout, int CoS) {
    *out = nullptr;
    auto* u = new UpSplit();
    u->Cs0 = Cs0; u->CoS = CoS;
    std::vector<uint16_t> wg((size_t)4 * CoS * Cs0, 0);
    for (int ab = 0; ab < 4; ++ab)
        for (int ci = 0; ci < Cin; ++ci)
            for (int co = 0; co < Cout; ++co)
                wg[((size_t)ab * CoS + co) * Cs0 + ci] = h_f2bf(w[((size_t)ab * Cin + ci) * Cout + co]);
    std::vector<float> bb(CoS, 0.0f);
    for (int c = 0; c < Cout; ++c) bb[c] = bias[c];
    auto bail = [&](int rc) { upsplit_free(u); return rc; };
    if (hipMalloc((void**)&u->d_w, wg.size() * 2) != hipSuccess || hipMalloc((void**)&u->d_bias, bb.size() * 4) != hipSuccess)
        return bail(fail(PSEG_ENOMEM, "hipMalloc(split up-conv weights) failed"));
    if (hipMemcpy(u->d_w, wg.data(), wg.size() * 2, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(u->d_bias, bb.data(), bb.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
        return bail(fail(PSEG_EHIP, "hipMemcpy(split up-conv weights) failed"));
    *out = u;
    return PSEG_OK;
}

// one thread per 16-byte (8-channel) output chunk; terms added in tap order (a, b) = 00, 01, 10, 11 on top of the bias
__global__ __launch_bounds__(256) void upsplit_sum_kernel(const uint16_t* __restrict__ D, int Hs, int Ws, int nch,
                                                          const float* __restrict__ bias, int relu, uint16_t* __restrict__ dst) {
    const size_t total = (size_t)4 * Hs * Ws * nch;
    const int Wo = 2 * Ws;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(t % nch);
        const size_t p = t / nch;
        const int x = (int)(p % Wo), y = (int)(p / Wo);
        float acc[8];
        const float4 b0 = *(const float4*)(bias + c8 * 8), b1 = *(const float4*)(bias + c8 * 8 + 4);
        acc[0] = b0.x; acc[1] = b0.y; acc[2] = b0.z; acc[3] = b0.w; acc[4] = b1.x; acc[5] = b1.y; acc[6] = b1.z; acc[7] = b1.w;
#pragma unroll
        for (int ab = 0; ab < 4; ++ab) {
            const int sy = (y + (ab >> 1)) >> 1, sx = (x + (ab & 1)) >> 1;
            if (sy >= Hs || sx >= Ws) continue;                       // 'same' padding of the k2 kernel: bottom / right zeros
            const uint4 v = *(const uint4*)(D + (((size_t)sy * Ws + sx) * 4 + ab) * (size_t)(nch * 8) + c8 * 8);
            const uint32_t q[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[2 * j] += __uint_as_float(q[j] << 16);
                acc[2 * j + 1] += __uint_as_float(q[j] & 0xffff0000u);
            }
        }
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a0 = acc[2 * j], a1 = acc[2 * j + 1];
            if (relu) { a0 = a0 > 0.f ? a0 : 0.f; a1 = a1 > 0.f ? a1 : 0.f; }
            typedef float f2 __attribute__((ext_vector_type(2)));
            typedef __bf16 b2 __attribute__((ext_vector_type(2)));
            o[j] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{a0, a1}, b2));
        }
        *(uint4*)(dst + p * (size_t)(nch * 8) + c8 * 8) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

int upsplit_launch(UpSplit* u, const uint16_t* src, int Hs, int Ws, uint16_t* dst, int relu, hipStream_t st) {
    const int64_t M = (int64_t)Hs * Ws, N = 4 * (int64_t)u->CoS, K = u->Cs0;
    if (M * K * 2 >= (int64_t)0xfffffff0u || N * K * 2 >= (int64_t)0xfffffff0u)
        return fail(PSEG_EUNSUPPORTED, "split up-conv operand larger than a 32-bit buffer range");
    if (!PSEG_KNOB("PSEG_UPSPLIT_TWO_PASS")) {
        // products kept in LDS, outputs written by the same workgroup (upsplit_fused_kernel)
        static bool fattr[64] = {false};
        int fdev = 0;
        PSEG_HIP(hipGetDevice(&fdev));
        if (!fattr[fdev & 63]) {
            PSEG_HIP(hipFuncSetAttribute((const void*)upsplit_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
            fattr[fdev & 63] = true;
        }
        // products in registers (upsplit_reg_kernel) where the GEMM is short, in LDS (upsplit_fused_kernel) where it is long -- same box,
        // 2048x1536 unet, registers / LDS: 128 -> 64 channels 200 / 215 us, 256 -> 128 141 / 141, 512 -> 256 103 / 96, 1024 -> 512 91 / 80
        // (the register form multiplies every other source row twice and reads 11 fragments per 24 MFMAs)
        if (K <= 128 && !PSEG_KNOB("PSEG_UPSPLIT_LDS")) {
            static bool rattr[64] = {false};
            if (!rattr[fdev & 63]) {
                PSEG_HIP(hipFuncSetAttribute((const void*)upsplit_reg_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * UR_STAGE));
                rattr[fdev & 63] = true;
            }
            const dim3 gr((unsigned)cdiv(u->CoS, 32), (unsigned)(cdiv(Hs, 8) * cdiv(Ws, 15)));
            upsplit_reg_kernel<<<gr, 256, 2 * UR_STAGE, st>>>(src, u->d_w, u->d_bias, dst, Hs, Ws, (int)K, u->CoS, relu);
            PSEG_HIP(hipGetLastError());
            return PSEG_OK;
        }
        const dim3 grid((unsigned)cdiv(u->CoS, 32), (unsigned)(cdiv(Hs, UF_TY - 1) * cdiv(Ws, UF_TX - 1)));
        upsplit_fused_kernel<<<grid, 256, 65536, st>>>(src, u->d_w, u->d_bias, dst, Hs, Ws, (int)K, u->CoS, relu);
        PSEG_HIP(hipGetLastError());
        return PSEG_OK;
    }
    const size_t need = (size_t)M * N * 2;
    if (need > u->D_bytes) {
        PSEG_HIP(hipStreamSynchronize(st));                            // the old buffer may still be read by a queued pass
        (void)hipFree(u->d_D);
        u->d_D = nullptr; u->D_bytes = 0;
        if (hipMalloc((void**)&u->d_D, need) != hipSuccess) return fail(PSEG_ENOMEM, "hipMalloc(%zu) for the split up-conv failed", need);
        u->D_bytes = need;
    }
    static bool attr_set[64] = {false};
    int dev = 0;
    PSEG_HIP(hipGetDevice(&dev));
    if (!attr_set[dev & 63]) {
        PSEG_HIP(hipFuncSetAttribute((const void*)upsplit_gemm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        attr_set[dev & 63] = true;
    }
    upsplit_gemm_kernel<<<dim3((unsigned)cdiv((int)M, GM), (unsigned)cdiv((int)N, GN)), 256, 65536, st>>>(src, u->d_w, u->d_D, (int)M, (int)N, (int)K);
    const int nch = u->CoS / 8;
    const size_t total = (size_t)4 * M * nch;
    upsplit_sum_kernel<<<(int)std::min<size_t>((total + 255) / 256, 16384), 256, 0, st>>>(u->d_D, Hs, Ws, nch, u->d_bias, relu, dst);
    PSEG_HIP(hipGetLastError());
    return PSEG_OK;
}

}  // namespace pseg
