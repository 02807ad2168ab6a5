// pseg_upsplit.hip -- UpSampling2D(2) -> Conv2D(k2, 'same') (unet, lib/model.py:174-175,180-181,186-187) in split form.
//
// out(y, x) = sum_{a,b} W[a][b] . src((y+a) >> 1, (x+b) >> 1): every tap of every output pixel is a product
// P_ab(Y, X) = W[a][b] . src(Y, X) of a SOURCE pixel, and each P_ab(Y, X) is used by (up to) four output pixels.
// Computing the four products once per source pixel is a plain GEMM  D[M][4 Cout] = src[M][Cin] . Wg^T  with a
// quarter of the direct form's MACs (Cin Cout per output pixel instead of 4 Cin Cout); the output is then a
// 4-term gather-sum over D (+ bias, ReLU), an HBM-bound pass.  The GEMM is a plain library GEMM (hipBLASLt, bf16
// operands, f32 accumulation, the "TN" layout: both operands K-contiguous); the partial products are stored as
// bf16, one more rounding than the direct kernel (same class as the per-layer activation rounding; the bf16
// parity tests cover it).  Used where the deep layers make the direct kernel weight-stream bound (Cin >= 256).
#include <hipblaslt/hipblaslt.h>

#include "pseg_common.h"

namespace pseg {

struct UpSplit {
    int Cs0 = 0, CoS = 0;
    uint16_t* d_w = nullptr;     // [4 CoS][Cs0] bf16, row n = (a*2+b) * CoS + co
    float* d_bias = nullptr;     // [CoS]
    uint16_t* d_D = nullptr;     // [M][4 CoS] bf16 partial products
    size_t D_bytes = 0;
    void* d_ws = nullptr;
    size_t ws_bytes = 32u << 20;
    hipblasLtMatmulDesc_t desc = nullptr;
    hipblasLtMatrixLayout_t lA = nullptr, lB = nullptr, lD = nullptr;
    hipblasLtMatmulHeuristicResult_t algo{};
    int64_t algo_M = -1;
};

static hipblasLtHandle_t lt_handle() {
    static thread_local hipblasLtHandle_t h[64] = {nullptr};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    if (!h[dev & 63] && hipblasLtCreate(&h[dev & 63]) != HIPBLAS_STATUS_SUCCESS) h[dev & 63] = nullptr;
    return h[dev & 63];
}

#define PSEG_LT(expr)                                                                             \
    do {                                                                                          \
        const hipblasStatus_t s_ = (expr);                                                        \
        if (s_ != HIPBLAS_STATUS_SUCCESS) return fail(PSEG_EHIP, "hipBLASLt: %s -> %d", #expr, (int)s_); \
    } while (0)

void upsplit_free(UpSplit* u) {
    if (!u) return;
    (void)hipFree(u->d_w); (void)hipFree(u->d_bias); (void)hipFree(u->d_D); (void)hipFree(u->d_ws);
    if (u->lA) hipblasLtMatrixLayoutDestroy(u->lA);
    if (u->lB) hipblasLtMatrixLayoutDestroy(u->lB);
    if (u->lD) hipblasLtMatrixLayoutDestroy(u->lD);
    if (u->desc) hipblasLtMatmulDescDestroy(u->desc);
    delete u;
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
    if (hipMalloc((void**)&u->d_w, wg.size() * 2) != hipSuccess || hipMalloc((void**)&u->d_bias, bb.size() * 4) != hipSuccess ||
        hipMalloc(&u->d_ws, u->ws_bytes) != hipSuccess)
        return bail(fail(PSEG_ENOMEM, "hipMalloc(split up-conv weights) failed"));
    if (hipMemcpy(u->d_w, wg.data(), wg.size() * 2, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(u->d_bias, bb.data(), bb.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
        return bail(fail(PSEG_EHIP, "hipMemcpy(split up-conv weights) failed"));
    if (hipblasLtMatmulDescCreate(&u->desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) != HIPBLAS_STATUS_SUCCESS)
        return bail(fail(PSEG_EHIP, "hipblasLtMatmulDescCreate failed"));
    const hipblasOperation_t opT = HIPBLAS_OP_T, opN = HIPBLAS_OP_N;
    if (hipblasLtMatmulDescSetAttribute(u->desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opT, sizeof(opT)) != HIPBLAS_STATUS_SUCCESS ||
        hipblasLtMatmulDescSetAttribute(u->desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opN, sizeof(opN)) != HIPBLAS_STATUS_SUCCESS)
        return bail(fail(PSEG_EHIP, "hipblasLtMatmulDescSetAttribute failed"));
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
    hipblasLtHandle_t h = lt_handle();
    if (!h) return fail(PSEG_EHIP, "hipblasLtCreate failed");
    const int64_t M = (int64_t)Hs * Ws, N = 4 * (int64_t)u->CoS, K = u->Cs0;
    const size_t need = (size_t)M * N * 2;
    if (need > u->D_bytes) {
        PSEG_HIP(hipStreamSynchronize(st));                            // the old buffer may still be read by a queued pass
        (void)hipFree(u->d_D);
        u->d_D = nullptr; u->D_bytes = 0;
        if (hipMalloc((void**)&u->d_D, need) != hipSuccess) return fail(PSEG_ENOMEM, "hipMalloc(%zu) for the split up-conv failed", need);
        u->D_bytes = need;
    }
    if (u->algo_M != M) {
        if (u->lA) hipblasLtMatrixLayoutDestroy(u->lA);
        if (u->lB) hipblasLtMatrixLayoutDestroy(u->lB);
        if (u->lD) hipblasLtMatrixLayoutDestroy(u->lD);
        u->lA = u->lB = u->lD = nullptr;
        // column-major view: D'(N x M) = op_T(Wg'(K x N)) . src'(K x M)
        PSEG_LT(hipblasLtMatrixLayoutCreate(&u->lA, HIP_R_16BF, K, N, K));
        PSEG_LT(hipblasLtMatrixLayoutCreate(&u->lB, HIP_R_16BF, K, M, K));
        PSEG_LT(hipblasLtMatrixLayoutCreate(&u->lD, HIP_R_16BF, N, M, N));
        hipblasLtMatmulPreference_t pref = nullptr;
        PSEG_LT(hipblasLtMatmulPreferenceCreate(&pref));
        hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &u->ws_bytes, sizeof(u->ws_bytes));
        int found = 0;
        const hipblasStatus_t hs = hipblasLtMatmulAlgoGetHeuristic(h, u->desc, u->lA, u->lB, u->lD, u->lD, pref, 1, &u->algo, &found);
        hipblasLtMatmulPreferenceDestroy(pref);
        if (hs != HIPBLAS_STATUS_SUCCESS || found < 1) return fail(PSEG_EUNSUPPORTED, "hipBLASLt has no bf16 GEMM for %lld x %lld x %lld", (long long)M, (long long)N, (long long)K);
        u->algo_M = M;
    }
    const float alpha = 1.0f, beta = 0.0f;
    PSEG_LT(hipblasLtMatmul(h, u->desc, &alpha, u->d_w, u->lA, src, u->lB, &beta, u->d_D, u->lD, u->d_D, u->lD, &u->algo.algo,
                            u->d_ws, u->ws_bytes, st));
    const int nch = u->CoS / 8;
    const size_t total = (size_t)4 * M * nch;
    upsplit_sum_kernel<<<(int)std::min<size_t>((total + 255) / 256, 16384), 256, 0, st>>>(u->d_D, Hs, Ws, nch, u->d_bias, relu, dst);
    PSEG_HIP(hipGetLastError());
    return PSEG_OK;
}

}  // namespace pseg
