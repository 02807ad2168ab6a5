// pseg_wgrad.h -- argument block and output helpers shared by the weight-gradient kernels (pseg_train.hip, pseg_wgrad_flat.hip).
#pragma once
#include "pseg_common.h"

namespace pseg {

// dW[tap][ci0+ci][co] += sum over the pixel strip of X[src pixel of (p, tap)][ci] * dY'[p][co]
// mode 0 (conv / logits): p = (y,x) of dY, X pixel = (y + ky - pt, x + kx - pl) (zero outside)
// mode 1 (deconv k2s2):   p = (i,j) of X,  dY pixel = (2i + a, 2j + b), tap = ab
struct WgradArgs {
    const float* X;
    int XC, ci0, Hx, Wx, xpitch;
    const float* dY;
    const float* maskY;
    int Hy, Wy, ypitch, Cout, Cin;
    int KW, pt, pl, mode, strip_rows;
    float* dW;
    float* dB;   // only written by blocks with tap 0 when non-null
    // matrix-core kernel only: conv stride (X pixel = y * stride + ky - pt), X stored at half resolution and
    // read through a nearest x2 upsample (Hx, Wx are then the upsampled extents, xpitch the stored row pitch),
    // pre-activation ReLU on X (res_unet)
    int stride = 1, xup = 0, in_relu = 0;
    // Deterministic form (default): a workgroup does not add its strip's partial sums into dW / dB with float atomics (whose
    // arrival order, and with it the last bits of the sum, changed from run to run) but stores them -- every element of its
    // (tap, channel block) exactly once -- into its own row of a scratch array, part[strip][tap][ci][co] (XC channels of this
    // launch) and partB[strip * 4 + wave][co]; wgrad_reduce_kernel then sums the strips in index order.
    float* part = nullptr;
    float* partB = nullptr;
    size_t pstride = 0;      // floats per strip row of `part` = taps * XC * Cout
    int cgroups = 1;         // flattened-row kernel: column groups a row strip is cut into (strip index = row strip * cgroups + group)
    // two-source kernel (wgrad_pair_kernel): XC = XC0 + XC1 channels of the concatenated input, the first XC0 from X, the rest from X1
    const float* X1 = nullptr;
    int XC0 = 0;
};

// one weight-gradient element / one bias partial of a strip leaves the kernel
__device__ __forceinline__ void wg_out(const WgradArgs& a, int strip, int tap, int ci, int co, float v) {
    if (a.part) a.part[(size_t)strip * a.pstride + ((size_t)tap * a.XC + ci) * a.Cout + co] = v;
    else atomicAdd(&a.dW[((size_t)tap * a.Cin + a.ci0 + ci) * a.Cout + co], v);
}
__device__ __forceinline__ void wg_out_bias(const WgradArgs& a, int strip, int wave, int co, float v) {
    if (a.partB) a.partB[((size_t)strip * 4 + wave) * a.Cout + co] = v;
    else atomicAdd(&a.dB[co], v);
}

// Flattened-row matrix-core weight gradient (pseg_wgrad_flat.hip): the instance table is keyed on (XC, Cout, KW); a layer
// with no instance stays on the kernels of pseg_train.hip.  plan() chooses the instance and the strips (= workgroups that
// each leave one row of `part`); launch() runs it with a.part / a.partB / a.pstride already set by the caller.
struct WgradFlatPlan {
    int instance = -1;       // index into the instance table
    int nstrips = 0;         // rows of `part` the launch writes (column groups x row strips)
    int strip_rows = 0, cgroups = 0;
};
bool wgrad_flat_plan(const WgradArgs& a, int taps, WgradFlatPlan* plan);
// ... and of the layers whose time is the tensors they read: the k2 s2 transposed convs and the 1x1 logits layer, both concat
// sources and the bias gradient in ONE pass over dY (a: X / XC0 = first source, X1 / XC - XC0 = second, dB set)
bool wgrad_pair_plan(const WgradArgs& a, int taps, WgradFlatPlan* plan);
int wgrad_pair_launch(const WgradArgs& a, const WgradFlatPlan& plan, hipStream_t st);
int wgrad_flat_launch(const WgradArgs& a, const WgradFlatPlan& plan, hipStream_t st);

}  // namespace pseg
