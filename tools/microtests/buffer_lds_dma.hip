// Micro-test (diagnostic, not product): semantics of buffer_load_dwordx4 ... lds on gfx950.
//  (1) a lane whose offset is out of the descriptor's range writes ZEROS to its LDS slot;
//  (2) an EXEC-masked lane writes nothing and active lanes keep slot = lane id.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned* src, unsigned nbytes, unsigned* out) {
    __shared__ __attribute__((aligned(16))) unsigned lds[64 * 4];
    const int lane = threadIdx.x;
    for (int i = 0; i < 4; ++i) lds[lane * 4 + i] = 0xdeadbeefu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    // lanes 0..31: in range; 32..47: out of range; 48..63: masked off
    unsigned voff = lane < 32 ? lane * 16 : 0x40000000u;
    if (lane < 48)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = 0; i < 4; ++i) out[lane * 4 + i] = lds[lane * 4 + i];
}
int main() {
    std::vector<unsigned> h(64 * 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 1000 + i;
    unsigned *d, *o;
    hipMalloc(&d, h.size() * 4); hipMalloc(&o, h.size() * 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, 32 * 16, o);
    std::vector<unsigned> r(64 * 4);
    hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost);
    int ok1 = 1, ok2 = 1, ok3 = 1;
    for (int l = 0; l < 32; ++l) for (int i = 0; i < 4; ++i) ok1 &= r[l * 4 + i] == 1000u + l * 4 + i;
    for (int l = 32; l < 48; ++l) for (int i = 0; i < 4; ++i) ok2 &= r[l * 4 + i] == 0u;
    for (int l = 48; l < 64; ++l) for (int i = 0; i < 4; ++i) ok3 &= r[l * 4 + i] == 0xdeadbeefu;
    printf("in-range copied: %d  OOB wrote zeros: %d (sample %08x)  masked untouched: %d (sample %08x)\n", ok1, ok2, r[32 * 4], ok3, r[48 * 4]);
    return !(ok1 && ok2 && ok3);
}
