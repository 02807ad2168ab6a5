// Micro-test (diagnostic, not product): v_mfma_f32_4x4x1_16b_f32 -- sixteen independent 4x4 outer-product blocks, K = 1.
//  (1) is D_blk[i][j] = fmaf(A_blk[i], B_blk[j], C_blk[i][j]) bitwise, so that a chain of them equals the sequential fmaf chain?
//  (2) what does one cost?  (a remainder pass for 4 / 8 left-over output channels wants it at a quarter / half of a 16-column tile)
// Layout: A: lane l holds A[blk = l >> 2][i = l & 3]; B: lane l holds B[blk = l >> 2][j = l & 3];
//         D: lane l holds D[blk = l >> 2][i = r][j = l & 3], r = 0..3 (the four registers are the four rows).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void k(const float* A, const float* B, const float* C, float* D, int K) {
    const int l = threadIdx.x;
    f32x4 acc;
    for (int r = 0; r < 4; ++r) acc[r] = C[((l >> 2) * 4 + r) * 4 + (l & 3)];
    for (int s = 0; s < K; ++s) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(A[s * 64 + l], B[s * 64 + l], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l >> 2) * 4 + r) * 4 + (l & 3)] = acc[r];
}
// throughput: N dependent-free MFMAs per wave (8 accumulators round robin), every SIMD of the chip busy
__global__ void t44(float* out, int n) {
    f32x4 acc[8];
    for (int q = 0; q < 8; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    for (int s = 0; s < n; ++s)
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[q], 0, 0, 0);
    float r = 0;
    for (int q = 0; q < 8; ++q) r += acc[q][0] + acc[q][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void t16(float* out, int n) {
    f32x4 acc[8];
    for (int q = 0; q < 8; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    for (int s = 0; s < n; ++s)
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[q], 0, 0, 0);
    float r = 0;
    for (int q = 0; q < 8; ++q) r += acc[q][0] + acc[q][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
int main() {
    const int K = 100;
    std::vector<float> A(K * 64), B(K * 64), C(256), D(256), R(256);
    srand(1);
    auto rnd = []() { return ((rand() % 20001) - 10000) / 3000.0f * ((rand() % 7) == 0 ? 1e-3f : 1.0f); };
    for (auto& v : A) v = rnd();
    for (auto& v : B) v = rnd();
    for (auto& v : C) v = rnd();
    for (int blk = 0; blk < 16; ++blk)
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                float acc = C[(blk * 4 + i) * 4 + j];
                for (int s = 0; s < K; ++s) acc = fmaf(A[s * 64 + blk * 4 + i], B[s * 64 + blk * 4 + j], acc);
                R[(blk * 4 + i) * 4 + j] = acc;
            }
    float *dA, *dB, *dC, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 1024); hipMalloc(&dD, 1024);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), 1024, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dB, dC, dD, K);
    hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    int same = 0; double maxd = 0;
    for (int i = 0; i < 256; ++i) { same += memcmp(&D[i], &R[i], 4) == 0; maxd = fmax(maxd, fabs((double)D[i] - R[i])); }
    printf("mfma_f32_4x4x1 vs fmaf chain (K=%d): %d / 256 bitwise equal, max abs diff %g\n", K, same, maxd);
    float* dO; hipMalloc(&dO, 1024 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int n = 20000;
    for (int which = 0; which < 2; ++which) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (which == 0) t44<<<1024, 256>>>(dO, n); else t16<<<1024, 256>>>(dO, n);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
        // 1024 blocks x 4 waves on 1024 SIMDs: 4 waves per SIMD in turn; MFMAs per SIMD = 4 * 8 * n
        const double per = ms * 1e-3 / (4.0 * 8 * n) * 2.1e9;
        printf("%s: %.3f ms, %.1f cycles per MFMA per SIMD at 2.1 GHz\n", which == 0 ? "4x4x1 " : "16x16x4", ms, per);
    }
    return same == 256 ? 0 : 1;
}
