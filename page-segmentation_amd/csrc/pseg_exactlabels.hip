// pseg_exactlabels.hip -- label-exact throughput mode (north_star: "label maps bit-identical to the CPU reference";
// lib/network.py:259 argmax on float32 logits).
//
// The bf16 engine's label map differs from the float32 engine's only where the two largest logits of a pixel are
// closer than the bf16 path's logit error.  This entry runs the bf16 graph with a margin output (top-1 minus top-2
// logit per pixel, written by the tail kernel), flags the pixels whose margin is below a threshold tau, and
// re-evaluates the flagged parts of the page with the float32 sequential-chain engine (the bit-exact referee, same
// weights): 64x64 blocks that hold a flagged pixel are covered by rectangles, each rectangle is cut out of the page
// with a 96-pixel halo (>= the 72-pixel receptive-field radius of fcn_skip, and a multiple of 32 so that pooling phase
// and the pad-to-32 canvas of an edge-touching crop equal the page's), predicted by the float32 companion, and its
// interior replaces the bf16 labels.  A float32 output pixel is one fmaf chain over its own receptive field, so a
// crop's interior pixels carry the same bits as the full page (tests: tiling invariance).
//
// tau is calibrated, not proven: tau0 = 4 x max |bf16 logit - float32 logit| over a calibration crop of the first page
// after a weight change, and every refereed rectangle is also a test -- it contains mostly UNflagged pixels, and an
// unflagged pixel whose float32 label differs from the bf16 one shows that tau is too small: tau doubles, the newly
// flagged blocks are refereed too, and the larger tau is kept for later pages.  When the flagged rectangles (with
// halos) approach the page's area, or tau keeps escalating, the whole page goes through the float32 engine.
#include <algorithm>
#include <cstring>

#include "pseg_common.h"

namespace pseg {

constexpr int XB = 64;        // flag block edge (pixels)
// crop halo: >= the receptive-field radius of the graph (fcn / fcn_skip: 75 pixels counting the one-sided growth of the
// 2x2 pools; unet ~122, res_unet ~124), a multiple of 32
static int halo_of(const Engine& e) { return (e.arch == PSEG_ARCH_FCN_SKIP || e.arch == PSEG_ARCH_FCN) ? 96 : 160; }

struct ExactState {
    pseg_engine* f32 = nullptr;          // float32 companion (PSEG_MODE_F32_EXACT, same graph and weights)
    float tau = 0.0f;                    // current threshold on the top-2 logit margin
    float calib_err = 0.0f;              // max |bf16 - float32| logit difference on the calibration crop
    float* d_margin = nullptr; size_t margin_bytes = 0;
    uint8_t* d_flags = nullptr; size_t flags_bytes = 0;
    float* d_blockmin = nullptr; size_t blockmin_bytes = 0;
    std::vector<float> h_blockmin;
    uint8_t* d_crop_img = nullptr; size_t crop_img_bytes = 0;
    uint8_t* d_crop_lab = nullptr; size_t crop_lab_bytes = 0;
    float* d_la = nullptr; float* d_lb = nullptr; size_t la_bytes = 0, lb_bytes = 0;   // calibration logits
    unsigned* d_counters = nullptr;      // [0] unflagged-but-different pixels, [1] flagged pixels, [2] max |dlogit| bits
    std::vector<uint8_t> h_flags, h_done;
    // statistics of the last call (pseg_label_exact_stats)
    double st_flag_px = 0, st_blocks = 0, st_area = 0, st_escal = 0, st_full = 0, st_rects = 0, st_changed = 0;
};

static int xensure(void** p, size_t* cap, size_t bytes) {
    if (*cap >= bytes && *p) return PSEG_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr; *cap = 0;
    PSEG_HIP(hipMalloc(p, bytes));
    *cap = bytes;
    return PSEG_OK;
}

void exact_free(Engine& e) {
    auto* x = (ExactState*)e.exact;
    if (!x) return;
    if (x->f32) (void)pseg_destroy(x->f32);
    (void)hipFree(x->d_margin); (void)hipFree(x->d_flags); (void)hipFree(x->d_blockmin); (void)hipFree(x->d_crop_img); (void)hipFree(x->d_crop_lab);
    (void)hipFree(x->d_la); (void)hipFree(x->d_lb); (void)hipFree(x->d_counters);
    delete x;
    e.exact = nullptr;
}

// flags[by][bx] = 1 when block (by, bx) holds a pixel with margin < tau; blockmin = the block's smallest margin;
// counters[1] += flagged pixels
__global__ __launch_bounds__(256) void flag_blocks_kernel(const float* margin, int H, int W, float tau, uint8_t* flags, float* blockmin,
                                                          int nbx, unsigned* counters) {
    const int bx = blockIdx.x, by = blockIdx.y;
    int cnt = 0;
    float mn = 3.4e38f;
    for (int i = threadIdx.x; i < XB * XB; i += 256) {
        const int y = by * XB + i / XB, x = bx * XB + i % XB;
        if (y < H && x < W) {
            const float m = margin[(size_t)y * W + x];
            mn = fminf(mn, m);
            if (m < tau) ++cnt;
        }
    }
    __shared__ int tot;
    __shared__ unsigned smin;
    if (threadIdx.x == 0) { tot = 0; smin = 0x7f7fffffu; }
    __syncthreads();
    if (cnt) atomicAdd(&tot, cnt);
    atomicMin(&smin, __float_as_uint(fmaxf(mn, 0.0f)));     // margins are >= 0: their bit patterns order as unsigned
    __syncthreads();
    if (threadIdx.x == 0) {
        flags[by * nbx + bx] = tot > 0;
        blockmin[by * nbx + bx] = __uint_as_float(smin);
        if (tot) atomicAdd(&counters[1], (unsigned)tot);
    }
}

// interior (oy0.., ox0.., h x w) of a refereed crop -> label map; counts pixels that change although their margin said "safe"
__global__ void referee_merge_kernel(const uint8_t* crop_lab, int crop_w, int cy0, int cx0, int oy0, int ox0, int h, int w,
                                     uint8_t* labels, int W, const float* margin, float tau, unsigned* counters) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= h * w) return;
    const int y = oy0 + i / w, x = ox0 + i % w;
    const uint8_t l32 = crop_lab[(size_t)(y - cy0) * crop_w + (x - cx0)];
    const size_t p = (size_t)y * W + x;
    if (l32 != labels[p]) {
        atomicAdd(&counters[3], 1u);
        const float m = margin[p];
        if (m >= tau) { atomicAdd(&counters[0], 1u); atomicMax(&counters[2], __float_as_uint(m)); }   // [2]: largest margin that still flipped
        labels[p] = l32;
    }
}

__global__ void max_abs_diff_kernel(const float* a, const float* b, size_t n, unsigned* counters) {
    float m = 0.0f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        m = fmaxf(m, fabsf(a[i] - b[i]));
    for (int sh = 32; sh > 0; sh >>= 1) m = fmaxf(m, __shfl_xor(m, sh));
    if ((threadIdx.x & 63) == 0) atomicMax(&counters[2], __float_as_uint(m));   // non-negative floats order as unsigned
}

__global__ void widen_u8_kernel(const uint8_t* in, int64_t* out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}

static int sync_weights(Engine& e, ExactState& x) {
    if (!x.f32) {
        PSEG_TRY(create_engine(e.arch, e.n_classes, e.in_ch, e.device, PSEG_MODE_F32_EXACT, e.flags, e.knobs, &x.f32));   // the companion reads its parent's knob snapshot
        e.exact_dirty = true;
    }
    if (!e.exact_dirty) return PSEG_OK;
    for (auto& p : e.params) {
        if (!p.set) return fail(PSEG_EINVAL, "weight '%s' was never set", p.name.c_str());
        PSEG_TRY(pseg_set_weights(x.f32, p.name.c_str(), p.host.data(), p.shape, p.ndim));
    }
    x.tau = 0.0f;   // recalibrate
    e.exact_dirty = false;
    return PSEG_OK;
}

// crop [y0, y1) x [x0, x1) of the page through the float32 companion; labels land in x.d_crop_lab (pitch x1 - x0)
static int referee_crop(Engine& e, ExactState& x, const uint8_t* d_img, int W, int y0, int x0, int y1, int x1, hipStream_t st,
                        float* d_logits = nullptr) {
    const int h = y1 - y0, w = x1 - x0;
    PSEG_TRY(xensure((void**)&x.d_crop_img, &x.crop_img_bytes, (size_t)h * w * e.in_ch));
    PSEG_TRY(xensure((void**)&x.d_crop_lab, &x.crop_lab_bytes, (size_t)h * w));
    PSEG_HIP(hipMemcpy2DAsync(x.d_crop_img, (size_t)w * e.in_ch, d_img + ((size_t)y0 * W + x0) * e.in_ch, (size_t)W * e.in_ch,
                              (size_t)w * e.in_ch, h, hipMemcpyDeviceToDevice, st));
    return predict_device(x.f32->e, x.d_crop_img, h, w, d_logits, nullptr, nullptr, x.d_crop_lab, st, nullptr);
}

static int calibrate(Engine& e, ExactState& x, const uint8_t* d_img, int H, int W, hipStream_t st) {
    // centre crop, 32-aligned origin, at most 512 x 512
    const int h = std::min(H, 512), w = std::min(W, 512);
    const int y0 = ((H - h) / 2) & ~31, x0 = ((W - w) / 2) & ~31;
    const size_t n = (size_t)h * w * e.n_classes;
    PSEG_TRY(xensure((void**)&x.d_la, &x.la_bytes, n * 4));
    PSEG_TRY(xensure((void**)&x.d_lb, &x.lb_bytes, n * 4));
    PSEG_TRY(referee_crop(e, x, d_img, W, y0, x0, y0 + h, x0 + w, st, x.d_la));
    PSEG_TRY(predict_device(e, x.d_crop_img, h, w, x.d_lb, nullptr, nullptr, nullptr, st, nullptr));   // the same crop as a page of its own
    PSEG_HIP(hipMemsetAsync(x.d_counters, 0, 16, st));
    max_abs_diff_kernel<<<(int)std::min<size_t>((n + 255) / 256, 2048), 256, 0, st>>>(x.d_la, x.d_lb, n, x.d_counters);
    unsigned c[4];
    PSEG_HIP(hipMemcpyAsync(c, x.d_counters, 16, hipMemcpyDeviceToHost, st));
    PSEG_HIP(hipStreamSynchronize(st));
    memcpy(&x.calib_err, &c[2], 4);
    x.tau = std::max(4.0f * x.calib_err, 1e-6f);
    if (const char* ev = PSEG_KNOB("PSEG_EXACT_TAU")) x.tau = (float)atof(ev);
    return PSEG_OK;
}

struct Rect { int by0, bx0, by1, bx1; };   // block coordinates, half open

// cover the to-do blocks with rectangles: horizontal runs per block row, runs with equal extent in consecutive rows merge
static std::vector<Rect> cover(const std::vector<uint8_t>& todo, int nby, int nbx) {
    std::vector<Rect> out, open;
    for (int by = 0; by <= nby; ++by) {
        std::vector<Rect> runs;
        if (by < nby)
            for (int bx = 0; bx < nbx;) {
                if (!todo[by * nbx + bx]) { ++bx; continue; }
                int b1 = bx;
                while (b1 < nbx && todo[by * nbx + b1]) ++b1;
                runs.push_back(Rect{by, bx, by + 1, b1});
                bx = b1;
            }
        std::vector<Rect> next;
        for (auto& r : runs) {
            bool merged = false;
            for (auto& o : open)
                if (o.by1 == by && o.bx0 == r.bx0 && o.bx1 == r.bx1 && (o.by1 - o.by0) < 8) { next.push_back(Rect{o.by0, o.bx0, by + 1, o.bx1}); o.by1 = -1; merged = true; break; }
            if (!merged) next.push_back(r);
        }
        for (auto& o : open) if (o.by1 >= 0) out.push_back(o);
        open.swap(next);
    }
    return out;
}

static int exact_labels(Engine& e, const uint8_t* d_img, int H, int W, uint8_t* d_labels_u8, int64_t* d_labels, float* d_margin_out,
                        hipStream_t st) {
    if (e.mode != PSEG_MODE_BF16) {   // a float32 engine is its own referee
        PSEG_TRY(predict_device(e, d_img, H, W, nullptr, nullptr, d_labels, d_labels_u8, st, d_margin_out));
        return PSEG_OK;
    }
    if (e.n_classes > 256) return fail(PSEG_EUNSUPPORTED, "label-exact mode keeps uint8 labels (<= 256 classes)");
    if (!e.exact) e.exact = new ExactState();
    ExactState& x = *(ExactState*)e.exact;
    PSEG_HIP(hipSetDevice(e.device));
    if (!x.d_counters) PSEG_HIP(hipMalloc((void**)&x.d_counters, 16));
    PSEG_TRY(sync_weights(e, x));
    const size_t npx = (size_t)H * W;
    float* d_margin = d_margin_out;
    if (!d_margin) { PSEG_TRY(xensure((void**)&x.d_margin, &x.margin_bytes, npx * 4)); d_margin = x.d_margin; }
    uint8_t* lab = d_labels_u8;
    x.st_flag_px = x.st_blocks = x.st_area = x.st_escal = x.st_full = x.st_rects = x.st_changed = 0;
    if (x.tau <= 0.0f) PSEG_TRY(calibrate(e, x, d_img, H, W, st));
    // 1. throughput pass with the margin map
    PSEG_TRY(predict_device(e, d_img, H, W, nullptr, nullptr, nullptr, lab, st, d_margin));
    const int nby = cdiv(H, XB), nbx = cdiv(W, XB), nblk = nby * nbx;
    PSEG_TRY(xensure((void**)&x.d_flags, &x.flags_bytes, (size_t)nblk));
    PSEG_TRY(xensure((void**)&x.d_blockmin, &x.blockmin_bytes, (size_t)nblk * 4));
    x.h_blockmin.assign(nblk, 0.0f);
    x.h_flags.assign(nblk, 0);
    x.h_done.assign(nblk, 0);
    bool full = false;
    double area = 0;
    const int XHALO = halo_of(e);
    for (int iter = 0; iter < 4 && !full; ++iter) {
        PSEG_HIP(hipMemsetAsync(x.d_counters, 0, 16, st));
        flag_blocks_kernel<<<dim3(nbx, nby), 256, 0, st>>>(d_margin, H, W, x.tau, x.d_flags, x.d_blockmin, nbx, x.d_counters);
        PSEG_HIP(hipMemcpyAsync(x.h_flags.data(), x.d_flags, nblk, hipMemcpyDeviceToHost, st));
        PSEG_HIP(hipMemcpyAsync(x.h_blockmin.data(), x.d_blockmin, (size_t)nblk * 4, hipMemcpyDeviceToHost, st));
        unsigned c[4];
        PSEG_HIP(hipMemcpyAsync(c, x.d_counters, 16, hipMemcpyDeviceToHost, st));
        PSEG_HIP(hipStreamSynchronize(st));
        x.st_flag_px = (double)c[1] / (double)npx;
        std::vector<uint8_t> todo(nblk);
        int ntodo = 0;
        for (int i = 0; i < nblk; ++i) { todo[i] = x.h_flags[i] && !x.h_done[i]; ntodo += todo[i]; }
        if (iter == 0) {
            // sentinels: the (up to four) unflagged blocks whose smallest margin is closest to tau go through the referee
            // too -- if tau is too small, this is where a flip shows
            for (int k = 0; k < 4; ++k) {
                int best = -1;
                for (int i = 0; i < nblk; ++i)
                    if (!todo[i] && !x.h_done[i] && (best < 0 || x.h_blockmin[i] < x.h_blockmin[best])) best = i;
                if (best < 0) break;
                todo[best] = 1;
                ++ntodo;
            }
        }
        if (!ntodo) break;
        const std::vector<Rect> rects = cover(todo, nby, nbx);
        double a = 0;
        for (auto& r : rects) {
            const int y0 = std::max(r.by0 * XB - XHALO, 0), x0 = std::max(r.bx0 * XB - XHALO, 0);
            const int y1 = std::min(r.by1 * XB + XHALO, H), x1 = std::min(r.bx1 * XB + XHALO, W);
            a += (double)(y1 - y0) * (x1 - x0);
        }
        if (area + a > 0.8 * (double)npx) { full = true; break; }
        area += a;
        for (auto& r : rects) {
            const int y0 = std::max(r.by0 * XB - XHALO, 0), x0 = std::max(r.bx0 * XB - XHALO, 0);
            const int y1 = std::min(r.by1 * XB + XHALO, H), x1 = std::min(r.bx1 * XB + XHALO, W);
            PSEG_TRY(referee_crop(e, x, d_img, W, y0, x0, y1, x1, st));
            const int oy0 = r.by0 * XB, ox0 = r.bx0 * XB, oh = std::min(r.by1 * XB, H) - oy0, ow = std::min(r.bx1 * XB, W) - ox0;
            referee_merge_kernel<<<cdiv(oh * ow, 256), 256, 0, st>>>(x.d_crop_lab, x1 - x0, y0, x0, oy0, ox0, oh, ow, lab, W, d_margin,
                                                                      x.tau, x.d_counters);
        }
        x.st_rects += (double)rects.size();
        for (int i = 0; i < nblk; ++i) x.h_done[i] |= todo[i];
        PSEG_HIP(hipMemcpyAsync(c, x.d_counters, 16, hipMemcpyDeviceToHost, st));
        PSEG_HIP(hipStreamSynchronize(st));
        x.st_changed += c[3];
        if (c[0] == 0) break;            // every label the referee changed had been flagged: tau held
        float worst;                     // an "unflagged" pixel flipped: the threshold was too small, also for later pages
        memcpy(&worst, &c[2], 4);
        x.tau = std::max(2.0f * x.tau, 2.0f * worst);
        x.st_escal += 1;
        if (iter == 3) full = true;
    }
    int ndone = 0;
    for (int i = 0; i < nblk; ++i) ndone += x.h_done[i];
    x.st_blocks = (double)ndone / nblk;
    x.st_area = area / (double)npx;
    if (full) {
        // near-ties everywhere (e.g. untrained weights): the referee takes the whole page
        PSEG_TRY(predict_device(x.f32->e, d_img, H, W, nullptr, nullptr, nullptr, lab, st, nullptr));
        x.st_full = 1;
        x.st_area = 1.0;
        x.st_blocks = 1.0;
    }
    if (d_labels) widen_u8_kernel<<<(int)std::min<size_t>((npx + 255) / 256, 8192), 256, 0, st>>>(lab, d_labels, npx);
    PSEG_HIP(hipGetLastError());
    return PSEG_OK;
}

}  // namespace pseg

using namespace pseg;

extern "C" {

int pseg_predict_margin_device(pseg_engine* h, const uint8_t* d_img, int H, int W, uint8_t* d_labels_u8, float* d_margin,
                               void* stream) {
    if (!h || !d_img || !d_margin) return fail(PSEG_EINVAL, "NULL argument");
    KnobScope knob_scope(h->e);
    hipStream_t st = stream ? (hipStream_t)stream : h->e.stream;
    return predict_device(h->e, d_img, H, W, nullptr, nullptr, nullptr, d_labels_u8, st, d_margin);
}

int pseg_predict_exact_labels_device(pseg_engine* h, const uint8_t* d_img, int H, int W, uint8_t* d_labels_u8,
                                     int64_t* d_labels, float* d_margin, void* stream) {
    if (!h || !d_img || !d_labels_u8) return fail(PSEG_EINVAL, "NULL argument (the uint8 label map is required)");
    KnobScope knob_scope(h->e);
    if (H <= 0 || W <= 0) return fail(PSEG_EINVAL, "empty page %dx%d", H, W);
    hipStream_t st = stream ? (hipStream_t)stream : h->e.stream;
    return exact_labels(h->e, d_img, H, W, d_labels_u8, d_labels, d_margin, st);
}

int pseg_predict_exact_labels(pseg_engine* h, const uint8_t* img, int H, int W, int64_t* labels, uint8_t* labels_u8) {
    if (!h || !img || (!labels && !labels_u8)) return fail(PSEG_EINVAL, "NULL argument");
    KnobScope knob_scope(h->e);
    if (H <= 0 || W <= 0) return fail(PSEG_EINVAL, "empty page %dx%d", H, W);
    Engine& e = h->e;
    PSEG_HIP(hipSetDevice(e.device));
    const size_t npx = (size_t)H * W;
    uint8_t* d_img = nullptr;
    PSEG_HIP(hipMalloc((void**)&d_img, npx * e.in_ch + npx + (labels ? npx * 8 : 0)));
    uint8_t* d_u8 = d_img + npx * e.in_ch;
    int64_t* d_i64 = labels ? (int64_t*)(d_img + ((npx * e.in_ch + npx + 7) & ~(size_t)7)) : nullptr;
    int rc = PSEG_OK;
    if (labels) {   // keep the int64 map 8-byte aligned inside the slab
        (void)hipFree(d_img);
        d_img = nullptr;
        const size_t off = (npx * e.in_ch + npx + 7) & ~(size_t)7;
        PSEG_HIP(hipMalloc((void**)&d_img, off + npx * 8));
        d_u8 = d_img + npx * e.in_ch;
        d_i64 = (int64_t*)(d_img + off);
    }
    if (hipMemcpyAsync(d_img, img, npx * e.in_ch, hipMemcpyHostToDevice, e.stream) != hipSuccess) rc = fail(PSEG_EHIP, "H2D copy failed");
    if (rc == PSEG_OK) rc = exact_labels(e, d_img, H, W, d_u8, d_i64, nullptr, e.stream);
    if (rc == PSEG_OK && labels_u8 && hipMemcpyAsync(labels_u8, d_u8, npx, hipMemcpyDeviceToHost, e.stream) != hipSuccess) rc = fail(PSEG_EHIP, "D2H copy failed");
    if (rc == PSEG_OK && labels && hipMemcpyAsync(labels, d_i64, npx * 8, hipMemcpyDeviceToHost, e.stream) != hipSuccess) rc = fail(PSEG_EHIP, "D2H copy failed");
    (void)hipStreamSynchronize(e.stream);
    (void)hipFree(d_img);
    return rc;
}

int pseg_label_exact_stats(const pseg_engine* h, double out[8]) {
    if (!h || !out) return fail(PSEG_EINVAL, "NULL argument");
    for (int i = 0; i < 8; ++i) out[i] = 0;
    const auto* x = (const ExactState*)h->e.exact;
    if (!x) return PSEG_OK;
    out[0] = x->tau; out[1] = x->calib_err; out[2] = x->st_flag_px; out[3] = x->st_blocks; out[4] = x->st_area;
    out[5] = x->st_escal; out[6] = x->st_full; out[7] = x->st_changed;
    return PSEG_OK;
}

}  // extern "C"
