// Micro-benchmark (diagnostic, not product): what one wave per SIMD sustains on the implicit-GEMM k-loop of the bf16 conv
// kernels -- MT x NT v_mfma_f32_16x16x32_bf16 per k-step with the next step's MT + NT ds_read_b128 fragment reads between
// them -- as a function of what else sits in the loop.  One 256-thread (or 512: a second, idle team) workgroup per CU.
//   V 0: MFMAs only            V 1: + fragment reads (64-byte pixel pitch, as the dense sigma = 4 tile)
//   V 2: + a counter read and a conditional branch per k-step (conv_sp_kernel's check)   V 3: V 1 with an 80-byte pitch
//   V 4: V 1 with the reads of step s + 2 (three register sets)
//   hipcc --offload-arch=gfx950 -O3 -o kloop.bin kloop.hip && ./kloop.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int NT, int V, int NW = 4, int MT = 4>
__global__ __launch_bounds__(NW * 64) void k(const unsigned short* src, int steps, int pitch, float* out, unsigned long long* tk) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 140 * 1024 / 16; i += NW * 64) ((uint4*)lds)[i] = ((const uint4*)src)[i];
    __syncthreads();
    const int p16 = lane & 15, g = lane >> 4;
    int pixbase[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) pixbase[m] = ((wave & 3) * 2 + (m >> 1)) * (36 * pitch + 16) + ((m & 1) * 16 + p16) * pitch + g * 16;
    const char* const ring = lds + 64 * 1024 + lane * 16;           // weights: 12 steps x NT KiB
    typedef __attribute__((address_space(3))) int lds_int;
    lds_int* const flag = (lds_int*)(lds + 139 * 1024);
    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 xa[MT], wa[NT], xb[MT], wb[NT], xc[MT], wc[NT];
#define LOAD(XF, WF, S)                                                                          \
    {                                                                                            \
        const int pos_ = (S) % 12, off_ = ((S) % 25) / 5 * (36 * pitch + 16) + ((S) % 5) * pitch; \
        WF[0] = *(const bf16x8*)(ring + pos_ * NT * 1024);                                       \
        _Pragma("unroll") for (int m = 0; m < MT; ++m) XF[m] = *(const bf16x8*)(lds + pixbase[m] + off_); \
        _Pragma("unroll") for (int t = 1; t < NT; ++t) WF[t] = *(const bf16x8*)(ring + pos_ * NT * 1024 + t * 1024); \
    }
#define MMA(XF, WF)                                                                              \
    _Pragma("unroll") for (int t = 0; t < NT; ++t)                                               \
        _Pragma("unroll") for (int m = 0; m < MT; ++m)                                           \
            acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WF[t], XF[m], acc[m][t], 0, 0, 0);
#define INTERLEAVE                                                                               \
    _Pragma("unroll") for (int q_ = 0; q_ < MT * NT; ++q_) {                                      \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                        \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                        \
        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);                                        \
    }
    LOAD(xa, wa, 0)
    if (V == 4) LOAD(xb, wb, 1)
    const long long t0 = __builtin_amdgcn_s_memtime();
    int s = 0;
    if constexpr (V == 4) {
        for (; s + 3 <= steps; s += 3) {
            __builtin_amdgcn_sched_barrier(0);
            LOAD(xc, wc, s + 2) MMA(xa, wa) INTERLEAVE
            __builtin_amdgcn_sched_barrier(0);
            LOAD(xa, wa, s + 3) MMA(xb, wb) INTERLEAVE
            __builtin_amdgcn_sched_barrier(0);
            LOAD(xb, wb, s + 4) MMA(xc, wc) INTERLEAVE
        }
    } else {
        for (; s + 2 <= steps; s += 2) {
            __builtin_amdgcn_sched_barrier(0);
            int f0 = 0;
            if (V == 2) f0 = flag[0];
            if (V != 0) LOAD(xb, wb, s + 1)
            MMA(xa, wa)
            INTERLEAVE
            __builtin_amdgcn_sched_barrier(0);
            if (V == 2) {
                if (__builtin_amdgcn_readfirstlane(f0) > s) { for (int it = 0; it < 4 && flag[1] > s; ++it) __builtin_amdgcn_s_sleep(1); }
                asm volatile("" ::: "memory");
            }
            int f1 = 0;
            if (V == 2) f1 = flag[0];
            if (V != 0) LOAD(xa, wa, s + 2)
            if constexpr (V != 0) { MMA(xb, wb) } else { MMA(xa, wa) }
            INTERLEAVE
            __builtin_amdgcn_sched_barrier(0);
            if (V == 2) {
                if (__builtin_amdgcn_readfirstlane(f1) > s) { for (int it = 0; it < 4 && flag[1] > s; ++it) __builtin_amdgcn_s_sleep(1); }
                asm volatile("" ::: "memory");
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) r += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
    out[blockIdx.x * (NW * 64) + tid] = r;
    if (tid == 0) tk[blockIdx.x] = (unsigned long long)(t1 - t0);
}

template <int NT, int V, int NW = 4, int MT = 4>
static void run(const char* what, const unsigned short* d, int pitch, float* dout, unsigned long long* dtk, int wgs) {
    const int steps = 600;
    hipFuncSetAttribute((const void*)k<NT, V, NW, MT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<NT, V, NW, MT><<<wgs, NW * 64, 140 * 1024>>>(d, 60, pitch, dout, dtk);
    hipEventRecord(a);
    k<NT, V, NW, MT><<<wgs, NW * 64, 140 * 1024>>>(d, steps, pitch, dout, dtk);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> h(wgs);
    hipMemcpy(h.data(), dtk, wgs * 8, hipMemcpyDeviceToHost);
    double tk = 0; for (auto v : h) tk += (double)v; tk /= wgs;
    const double per = tk / steps;
    printf("MT %d NT %d  waves/SIMD %d  %-46s WGs %3d: %7.1f ticks / k-step of a wave (%5.1f SIMD cycles per MFMA)  kernel %.3f ms  -> %.2f GHz  %.0f TFLOP/s\n", MT, NT, NW / 4, what, wgs, per,
           per / (MT * NT) / (NW / 4), ms, tk / (ms * 1e-3) * 1e-9, (double)wgs * NW * steps * MT * NT * 16384.0 / (ms * 1e-3) * 1e-12);
}
int main(int argc, char** argv) {
    const bool zeros = argc > 1 && atoi(argv[1]) == 0;
    std::vector<unsigned short> h(140 * 1024 / 2);
    srand(1);
    for (auto& v : h) v = zeros ? 0 : (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));   // random bf16 in +-[0.0078, 0.0156)
    unsigned short* d; float* dout; unsigned long long* dtk;
    hipMalloc(&d, h.size() * 2); hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMalloc(&dout, 256 * 768 * 4); hipMalloc(&dtk, 256 * 8); hipMemset(d + 139 * 512, 0, 64);
    printf("operands: %s\n", zeros ? "zeros" : "random");
    for (int wgs : {256}) {
        // register tiles: fragment reads per MFMA = 1 / MT + 1 / NT
        run<4, 1, 4, 4>("MT 4: + fragment reads", d, 64, dout, dtk, wgs);
        run<4, 1, 8, 4>("MT 4: + fragment reads", d, 64, dout, dtk, wgs);
        run<4, 1, 4, 8>("MT 8: + fragment reads", d, 64, dout, dtk, wgs);
        run<4, 1, 8, 8>("MT 8: + fragment reads", d, 64, dout, dtk, wgs);
        run<2, 1, 4, 8>("MT 8 NT 2 (conv2): + fragment reads", d, 64, dout, dtk, wgs);
        run<2, 1, 8, 8>("MT 8 NT 2 (conv2): + fragment reads", d, 64, dout, dtk, wgs);
        run<3, 1, 4, 8>("MT 8 NT 3: + fragment reads", d, 64, dout, dtk, wgs);
        run<3, 1, 8, 8>("MT 8 NT 3: + fragment reads", d, 64, dout, dtk, wgs);
        run<4, 1, 8, 3>("MT 3 NT 4 (8 x 24 tiles): + fragment reads", d, 64, dout, dtk, wgs);
        run<5, 1, 4, 3>("MT 3 NT 5 (8 x 24 tiles): + fragment reads", d, 64, dout, dtk, wgs);
        run<4, 0, 8, 4>("MFMAs only", d, 64, dout, dtk, wgs);
    }
    return 0;
}
