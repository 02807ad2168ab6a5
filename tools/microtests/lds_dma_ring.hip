// Micro-benchmark (diagnostic, not product): LDS-DMA fill rate of ONE workgroup per CU that streams a shared, L2-hot
// weight set (256 KiB, every workgroup the same bytes in the same order -- the 1/8-resolution k5 layers of fcn_skip)
// through an LDS ring, as a function of the number of issuing waves and of the requests each keeps in flight
// (continuous issue with counted vmcnt waits, not issue-all-then-drain as lds_dma_bw.hip).
//   hipcc --offload-arch=gfx950 -O3 -o lds_dma_ring.bin lds_dma_ring.hip && ./lds_dma_ring.bin
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NW, int Q>
__global__ __launch_bounds__(NW * 64) void k(const char* src, int span_kb, int rounds, unsigned long long* out) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pieces = span_kb;                       // 1 KiB pieces of the stream
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rounds; ++r) {
        int slot = 0;
        for (int p = wave; p < pieces; p += NW) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)p * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void*)(lds + (wave * Q + slot) * 1024), 16, 0, 0);
            slot = slot + 1 == Q ? 0 : slot + 1;
            // keep Q - 1 younger requests in flight
            if constexpr (Q == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if constexpr (Q == 2) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else if constexpr (Q == 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else if constexpr (Q == 8) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (unsigned long long)(t1 - t0);
    if (threadIdx.x == 1 && lds[5] == 77) out[0] = 1;
}
template <int NW, int Q>
static void run(const char* d, int span_kb, int wgs, unsigned long long* dout) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int lds = NW * Q * 1024, rounds = 20;
    hipFuncSetAttribute((const void*)k<NW, Q>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    k<NW, Q><<<wgs, NW * 64, lds>>>(d, span_kb, 2, dout);
    hipEventRecord(a);
    k<NW, Q><<<wgs, NW * 64, lds>>>(d, span_kb, rounds, dout);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    unsigned long long h[1024];
    hipMemcpy(h, dout, wgs * 8, hipMemcpyDeviceToHost);
    double tk = 0; for (int i = 0; i < wgs; ++i) tk += (double)h[i]; tk /= wgs;
    const double bytes_wg = (double)rounds * span_kb * 1024.0;
    printf("waves %2d  in flight/wave %2d  WGs %3d  span %3d KiB: %.3f ms  %6.1f GB/s chip  %5.1f B per 100 MHz tick per WG  (%.0f ticks per WG per pass)\n",
           NW, Q, wgs, span_kb, ms, bytes_wg * wgs / ms * 1e-6, bytes_wg / tk, tk / rounds);
}
int main() {
    char* d; unsigned long long* dout;
    hipMalloc(&d, 1 << 20); hipMemset(d, 1, 1 << 20); hipMalloc(&dout, 1024 * 8);
    for (int wgs : {192, 256}) {
        run<4, 2>(d, 256, wgs, dout); run<4, 4>(d, 256, wgs, dout); run<4, 8>(d, 256, wgs, dout); run<4, 16>(d, 256, wgs, dout);
        run<8, 2>(d, 256, wgs, dout); run<8, 4>(d, 256, wgs, dout); run<8, 8>(d, 256, wgs, dout);
        run<12, 4>(d, 256, wgs, dout); run<12, 8>(d, 256, wgs, dout);
        run<16, 2>(d, 256, wgs, dout); run<16, 4>(d, 256, wgs, dout); run<16, 8>(d, 256, wgs, dout);
    }
    return 0;
}
