/*
 * pseg_oracle.c -- CPU restatement of the ocr4all_pixel_classifier per-pixel hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
 * as the checker.  The product path (page-segmentation_amd/csrc) never links or calls it.
 *
 * PARITY UNPINNED (layer arithmetic): the reference ships no tests, no golden vectors and
 * no trained model, and its arithmetic lives in TensorFlow 2.5.0 / OpenCV 4.5.5, neither of
 * which is installed or installable here.  What IS pinned: the
 * reference's own importable code (lib/util.py, lib/architecture.py: preprocess LUT,
 * image_to_batch) and its one runnable third-party call (scipy.special.softmax) -- see
 * tests/golden/make_golden.py.  The layer arithmetic below is a restatement of published
 * TF/Keras semantics, cross-checked against torch.nn.functional on CPU (independent
 * implementation) in tests/test_oracle.py.
 *
 * Semantics restated (reference file:line):
 *   orc_preprocess      lib/architecture.py:67-68  (x / 255.0, cast to f32 by Keras)
 *   orc_conv2d          lib/model.py:50-66,88      (Conv2D, SAME, NHWC, bias, optional ReLU)
 *                       lib/model.py:69,75         (Conv2DTranspose k5 s1 == correlation with the
 *                                                   flipped kernel; caller passes correlation-form
 *                                                   weights)
 *   orc_deconv2x2       lib/model.py:71,79,83      (Conv2DTranspose k2 s2 SAME: non-overlapping
 *                                                   scatter of four 1x1 products)
 *   orc_maxpool2        lib/model.py:54,59,64      (MaxPooling2D 2x2 s2; dims even after pad-to-32)
 *   orc_round_bf16      build-defined: bf16 activation mode (round-to-nearest-even)
 *
 * Accumulation order (this is what the f32 "exact" HIP path reproduces bit for bit):
 *   acc = +0;
 *   for cb in 0, 16, 32, ... (blocks of ORC_CHAIN_BLOCK = 16 input channels of the -- concatenated -- input):
 *     for ky, for kx, for ci in the block (ascending): acc = fmaf(x, w, acc);
 *   out = acc + bias;  ReLU = max(out, 0).
 *   Out-of-image taps are skipped (== adding an exact zero product to an accumulator that can never be -0).
 * TensorFlow's own float32 summation order is unspecified (Eigen / oneDNN block and vectorise the contraction, and differ
 * between builds), so ANY fixed order is as faithful to "TF-CPU float32" as any other; this one is chosen because it is
 * what an LDS-tiled matrix-core kernel can follow at speed (a 16-channel slab of the halo tile per pass: round 3; rounds
 * 1-2 ran the chain over all channels inside each tap, which forced all-channel tiles).  A layer with <= 16 input
 * channels, every 1x1 convolution and the k2 s2 transposed convolution (one tap per output) are unchanged by the blocking.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_MAX_COUT 1024
#define ORC_CHAIN_BLOCK 16   /* input channels per pass of the accumulation chain (see the header) */

int orc_chain_block(void) { return ORC_CHAIN_BLOCK; }

int orc_abi_version(void) { return 1; }

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* lib/architecture.py:67-68 -- float32(float64(u) / 255.0) == float32(u) / 255.0f for all 256
 * byte values (checked in tests/test_oracle.py against the reference function itself). */
void orc_preprocess(const uint8_t* in, int64_t n, float* out) {
    for (int64_t i = 0; i < n; ++i) out[i] = (float)in[i] / 255.0f;
}

static inline float bf16_round(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return f; /* NaN stays NaN */
    u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
    memcpy(&f, &u, 4);
    return f;
}

void orc_round_bf16(float* buf, int64_t n) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) buf[i] = bf16_round(buf[i]);
}

/*
 * NHWC correlation.  w is [KH][KW][Cin][Cout] (Keras Conv2D kernel layout).
 * Input pixel for output (y,x), tap (ky,kx): (y*stride + ky - pt, x*stride + kx - pl).
 * TF SAME: pad_total = max((out-1)*stride + k - in, 0); pt = pad_total / 2 (floor) -- the
 * caller computes pt/pl so that even kernels / strided convs pad "after" (lib/model.py:174,281).
 */
int orc_conv2d(const float* in, int H, int W, int Cin, const float* w, const float* bias, int KH,
               int KW, int stride, int pt, int pl, int Hout, int Wout, int Cout, int relu,
               float* out) {
    if (Cout > ORC_MAX_COUT || Cout <= 0) return -1;
#pragma omp parallel for schedule(dynamic, 1)
    for (int y = 0; y < Hout; ++y) {
        float acc[ORC_MAX_COUT];
        for (int x = 0; x < Wout; ++x) {
            for (int co = 0; co < Cout; ++co) acc[co] = 0.0f;
            for (int cb = 0; cb < Cin; cb += ORC_CHAIN_BLOCK) {
                const int ce = cb + ORC_CHAIN_BLOCK < Cin ? cb + ORC_CHAIN_BLOCK : Cin;
                for (int ky = 0; ky < KH; ++ky) {
                    const int iy = y * stride + ky - pt;
                    if (iy < 0 || iy >= H) continue;
                    for (int kx = 0; kx < KW; ++kx) {
                        const int ix = x * stride + kx - pl;
                        if (ix < 0 || ix >= W) continue;
                        const float* px = in + ((int64_t)iy * W + ix) * Cin;
                        const float* wt = w + (int64_t)(ky * KW + kx) * Cin * Cout;
                        for (int ci = cb; ci < ce; ++ci) {
                            const float xv = px[ci];
                            const float* wr = wt + (int64_t)ci * Cout;
                            for (int co = 0; co < Cout; ++co)
                                acc[co] = __builtin_fmaf(xv, wr[co], acc[co]);
                        }
                    }
                }
            }
            float* o = out + ((int64_t)y * Wout + x) * Cout;
            for (int co = 0; co < Cout; ++co) {
                float v = acc[co] + (bias ? bias[co] : 0.0f);
                if (relu) v = v > 0.0f ? v : 0.0f;
                o[co] = v;
            }
        }
    }
    return 0;
}

/*
 * Conv2DTranspose k2 s2 SAME (lib/model.py:71,79,83):
 *   out[2i+a, 2j+b, co] = bias[co] + sum_ci in[i,j,ci] * w[a][b][ci][co]
 * w is passed as [2][2][Cin][Cout] (the caller transposes Keras' (kh,kw,Cout,Cin)).
 */
int orc_deconv2x2(const float* in, int H, int W, int Cin, const float* w, const float* bias,
                  int Cout, int relu, float* out) {
    if (Cout > ORC_MAX_COUT || Cout <= 0) return -1;
    const int Wo = 2 * W;
#pragma omp parallel for schedule(dynamic, 1)
    for (int i = 0; i < H; ++i) {
        float acc[ORC_MAX_COUT];
        for (int j = 0; j < W; ++j) {
            const float* px = in + ((int64_t)i * W + j) * Cin;
            for (int a = 0; a < 2; ++a)
                for (int b = 0; b < 2; ++b) {
                    const float* wt = w + (int64_t)(a * 2 + b) * Cin * Cout;
                    for (int co = 0; co < Cout; ++co) acc[co] = 0.0f;
                    for (int ci = 0; ci < Cin; ++ci) {
                        const float xv = px[ci];
                        const float* wr = wt + (int64_t)ci * Cout;
                        for (int co = 0; co < Cout; ++co)
                            acc[co] = __builtin_fmaf(xv, wr[co], acc[co]);
                    }
                    float* o = out + ((int64_t)(2 * i + a) * Wo + (2 * j + b)) * Cout;
                    for (int co = 0; co < Cout; ++co) {
                        float v = acc[co] + (bias ? bias[co] : 0.0f);
                        if (relu) v = v > 0.0f ? v : 0.0f;
                        o[co] = v;
                    }
                }
        }
    }
    return 0;
}

/* MaxPooling2D((2,2),(2,2)) on even dims (lib/model.py:54,59,64). */
int orc_maxpool2(const float* in, int H, int W, int C, float* out) {
    if ((H & 1) || (W & 1)) return -1;
    const int Ho = H / 2, Wo = W / 2;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < Ho; ++y)
        for (int x = 0; x < Wo; ++x)
            for (int c = 0; c < C; ++c) {
                const float a = in[((int64_t)(2 * y) * W + 2 * x) * C + c];
                const float b = in[((int64_t)(2 * y) * W + 2 * x + 1) * C + c];
                const float d = in[((int64_t)(2 * y + 1) * W + 2 * x) * C + c];
                const float e = in[((int64_t)(2 * y + 1) * W + 2 * x + 1) * C + c];
                float m = a > b ? a : b;
                const float n = d > e ? d : e;
                m = m > n ? m : n;
                out[((int64_t)y * Wo + x) * C + c] = m;
            }
    return 0;
}

/* np.argmax(logit, -1) (lib/network.py:259): first maximum wins. */
void orc_argmax(const float* logits, int64_t n, int C, int64_t* labels) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const float* z = logits + i * C;
        int best = 0;
        float bv = z[0];
        for (int c = 1; c < C; ++c)
            if (z[c] > bv) { bv = z[c]; best = c; }
        labels[i] = best;
    }
}
