// Micro-benchmark (diagnostic, not product): sustained LDS-DMA (global_load_lds_dwordx4) fill rate per CU
// on gfx950 when the source is (a) a small L2-resident block read by every workgroup (the packed conv
// weights) and (b) a large streamed buffer.  Prints bytes per clock per CU at 2.4 GHz nominal.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int INFLIGHT>
__global__ __launch_bounds__(256) void k(const char* src, size_t span, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    size_t off = ((size_t)blockIdx.x * 7919u * 4096u) % span;   // per-WG start (L2 block: span small -> all overlap)
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < INFLIGHT; ++j) {
            const char* s = src + ((off + (size_t)(j * 4 + wave) * 1024) % span) + lane * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s,
                                             (__attribute__((address_space(3))) void*)(lds + (j * 4 + wave) * 1024), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        off = (off + INFLIGHT * 4096) % span;
    }
    __syncthreads();
    if (threadIdx.x == 0 && lds[5] == 77) sink[0] = 1;
}
// same traffic through VGPRs: global_load_dwordx4 then ds_write_b128
template <int INFLIGHT>
__global__ __launch_bounds__(256) void kv(const char* src, size_t span, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    size_t off = ((size_t)blockIdx.x * 7919u * 4096u) % span;
    for (int it = 0; it < iters; ++it) {
        uint4 v[INFLIGHT];
#pragma unroll
        for (int j = 0; j < INFLIGHT; ++j)
            v[j] = *(const uint4*)(src + ((off + (size_t)(j * 4 + wave) * 1024) % span) + lane * 16);
#pragma unroll
        for (int j = 0; j < INFLIGHT; ++j) *(uint4*)(lds + (j * 4 + wave) * 1024 + lane * 16) = v[j];
        off = (off + INFLIGHT * 4096) % span;
    }
    __syncthreads();
    if (threadIdx.x == 0 && lds[5] == 77) sink[0] = 1;
}
template <int INFLIGHT>
static void runv(const char* name, const char* d, size_t span, int wgs, int iters, unsigned* sink) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int lds = INFLIGHT * 4096;
    hipFuncSetAttribute((const void*)kv<INFLIGHT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    kv<INFLIGHT><<<wgs, 256, lds>>>(d, span, 10, sink);
    hipEventRecord(a);
    kv<INFLIGHT><<<wgs, 256, lds>>>(d, span, iters, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)wgs * iters * INFLIGHT * 4096.0;
    printf("%-28s inflight/wave %2d  WGs %4d: %.3f ms  %.1f GB/s  %.1f B/clk/CU (256 CUs, 2.4 GHz)\n", name, INFLIGHT, wgs, ms,
           bytes / ms * 1e-6, bytes / (ms * 1e-3) / 256 / 2.4e9);
}
template <int INFLIGHT>
static void run(const char* name, const char* d, size_t span, int wgs, int iters, unsigned* sink) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int lds = INFLIGHT * 4096;
    hipFuncSetAttribute((const void*)k<INFLIGHT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    k<INFLIGHT><<<wgs, 256, lds>>>(d, span, 10, sink);
    hipEventRecord(a);
    k<INFLIGHT><<<wgs, 256, lds>>>(d, span, iters, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)wgs * iters * INFLIGHT * 4096.0;
    printf("%-28s inflight/wave %2d  WGs %4d: %.3f ms  %.1f GB/s  %.1f B/clk/CU (256 CUs, 2.4 GHz)\n", name, INFLIGHT, wgs, ms,
           bytes / ms * 1e-6, bytes / (ms * 1e-3) / 256 / 2.4e9);
}
int main() {
    char* d; unsigned* sink;
    const size_t big = 1ull << 30;
    hipMalloc(&d, big); hipMemset(d, 1, big); hipMalloc(&sink, 4);
    for (int wgs : {256, 512, 1024}) {
        run<2>("L2-resident 64 KiB", d, 64 * 1024, wgs, 400, sink);
        run<4>("L2-resident 64 KiB", d, 64 * 1024, wgs, 400, sink);
        run<8>("L2-resident 64 KiB", d, 64 * 1024, wgs, 400, sink);
        run<8>("streamed 1 GiB", d, big, wgs, 400, sink);
        runv<4>("VGPR path, L2-resident", d, 64 * 1024, wgs, 400, sink);
        runv<8>("VGPR path, L2-resident", d, 64 * 1024, wgs, 400, sink);
        runv<8>("VGPR path, streamed", d, big, wgs, 400, sink);
    }
    return 0;
}
