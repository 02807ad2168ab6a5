// pseg_exactlabels.hip -- label-exact throughput mode (north_star: "label maps bit-identical to the CPU reference";
// lib/network.py:259 argmax on float32 logits).
//
// The bf16 engine's label map differs from the float32 engine's only where the two largest logits of a pixel are
// closer than the bf16 path's error on their difference.  This entry runs the bf16 graph with a margin output (top-1
// minus top-2 logit per pixel, written by the tail kernel), flags the pixels whose margin is below a threshold tau, and
// re-evaluates the flagged parts of the page with the float32 engine (the bit-exact referee, same weights): 32x32
// blocks that hold a flagged pixel are grouped into rectangles by a cost model (blocks whose halos overlap share a
// crop; two crops merge when one launch over their bounding box is cheaper than two launches), each rectangle is cut
// out of the page with a 96-pixel halo (>= the 75-pixel receptive-field radius of fcn_skip, and a multiple of 32 so
// that pooling phase and the pad-to-32 canvas of an edge-touching crop equal the page's), predicted by the float32
// companion, and its interior replaces the bf16 labels.  A float32 output pixel is one fmaf chain over its own
// receptive field, so a crop's interior pixels carry the same bits as the full page (tests: tiling invariance).  When
// the cost model prices the crops above one float32 pass over the whole page, the page goes through the float32
// engine whole (pages whose class boundaries run through every block: text pages at their line pitch).
//
// tau is CALIBRATED, NOT PROVEN -- pseg_predict (PSEG_MODE_F32_EXACT) is the only mode that is bit-exact by
// construction.  What keeps tau honest: (1) after a weight change it starts from 4 x max |bf16 logit - float32 logit| over
// three calibration crops of the page (quarter points and centre); (2) every refereed rectangle is a measurement: the
// companion also returns ITS margin map, and for every refereed pixel the change of the margin (same label: |m_bf16 -
// m_f32|; another label: m_bf16 + m_f32 -- the two logits moved past each other by at least that much) is folded into a
// running maximum E that lives across pages until the weights change: tau >= 2 E always, and a grown tau re-flags the
// page at once; (3) the (up to four) unflagged blocks with the smallest margins are refereed as sentinels on every
// page; (4) an unflagged pixel that flips inside a refereed rectangle doubles tau.  An unflagged pixel outside every
// refereed rectangle is trusted on that evidence.
#include <algorithm>
#include <cstring>

#include "pseg_common.h"

namespace pseg {

constexpr int XB = 32;        // flag block edge (pixels): a multiple of the graphs' pad unit, so crops start on the page's 32-pixel grid
// crop halo: >= the receptive-field radius of the graph (fcn / fcn_skip: 75 pixels counting the one-sided growth of the
// 2x2 pools; unet ~122, res_unet ~124), a multiple of 32
static int halo_of(const Engine& e) { return (e.arch == PSEG_ARCH_FCN_SKIP || e.arch == PSEG_ARCH_FCN) ? 96 : 160; }

struct ExactState {
    pseg_engine* f32 = nullptr;          // float32 companion (PSEG_MODE_F32_EXACT, same graph and weights)
    float tau = 0.0f;                    // current threshold on the top-2 logit margin
    float calib_err = 0.0f;              // max |bf16 - float32| logit difference on the calibration crops
    float margin_err = 0.0f;             // running max change of the top-2 margin between the bf16 pass and the referee (all refereed pixels since the last weight change)
    int calib_Hp = 0, calib_Wp = 0;      // canvas the calibration crops were taken from
    int pages_since_calib = 0;
    int fallback_streak = 0, skip_left = 0;   // pages in a row that went through the float32 engine whole / pages left to send there directly
    int skip_len = 0;                         // length of the last direct stretch: doubles while the probes behind it keep falling back
    float* d_margin = nullptr; size_t margin_bytes = 0;
    uint8_t* d_flags = nullptr; size_t flags_bytes = 0;
    float* d_blockmin = nullptr; size_t blockmin_bytes = 0;
    std::vector<float> h_blockmin;
    uint8_t* d_crop_img = nullptr; size_t crop_img_bytes = 0;
    uint8_t* d_crop_lab = nullptr; size_t crop_lab_bytes = 0;
    float* d_crop_logits = nullptr; size_t crop_logits_bytes = 0;   // the companion's logits of the crop ...
    float* d_crop_margin = nullptr; size_t crop_margin_bytes = 0;   // ... and its top-2 margin
    uint8_t* d_page = nullptr; size_t page_bytes = 0;               // host-buffer entry: page + label maps (persistent workspace)
    float* d_la = nullptr; float* d_lb = nullptr; size_t la_bytes = 0, lb_bytes = 0;   // calibration logits
    unsigned* d_counters = nullptr;      // [0] unflagged-but-different pixels, [1] flagged pixels, [2] max |dlogit| / largest flipped margin bits, [3] labels changed, [4] max margin change bits
    std::vector<uint8_t> h_flags, h_done;
    // statistics of the last call (pseg_label_exact_stats)
    double st_flag_px = 0, st_blocks = 0, st_area = 0, st_escal = 0, st_full = 0, st_rects = 0, st_changed = 0, st_cost = 0, st_direct = 0;
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
    (void)hipFree(x->d_la); (void)hipFree(x->d_lb); (void)hipFree(x->d_counters); (void)hipFree(x->d_crop_logits); (void)hipFree(x->d_crop_margin); (void)hipFree(x->d_page);
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

// interior (oy0.., ox0.., h x w) of a refereed crop -> label map; counts pixels that change although their margin said
// "safe", and folds the change of every refereed pixel's margin into counters[4] (the running estimate tau rests on)
__global__ void referee_merge_kernel(const uint8_t* crop_lab, const float* crop_margin, int crop_w, int cy0, int cx0, int oy0, int ox0,
                                     int h, int w, uint8_t* labels, int W, const float* margin, float tau, unsigned* counters) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float dm = 0.0f;
    if (i < h * w) {
        const int y = oy0 + i / w, x = ox0 + i % w;
        const size_t cp = (size_t)(y - cy0) * crop_w + (x - cx0);
        const uint8_t l32 = crop_lab[cp];
        const float mf = crop_margin[cp];
        const size_t p = (size_t)y * W + x;
        const float m = margin[p];
        if (l32 != labels[p]) {
            atomicAdd(&counters[3], 1u);
            if (m >= tau) { atomicAdd(&counters[0], 1u); atomicMax(&counters[2], __float_as_uint(m)); }   // [2]: largest margin that still flipped
            labels[p] = l32;
            dm = m + mf;            // the two logits passed each other: their difference moved by at least this much
        } else {
            dm = fabsf(m - mf);
        }
        if (!(dm < 3.0e38f)) dm = 0.0f;   // one-class graphs carry +inf margins
    }
    for (int sh = 32; sh > 0; sh >>= 1) dm = fmaxf(dm, __shfl_xor(dm, sh));
    if ((threadIdx.x & 63) == 0 && dm > 0.0f) atomicMax(&counters[4], __float_as_uint(dm));
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
    x.margin_err = 0.0f;
    x.fallback_streak = x.skip_left = x.skip_len = 0;
    e.exact_dirty = false;
    return PSEG_OK;
}

// crop [y0, y1) x [x0, x1) of the page through the float32 companion; labels land in x.d_crop_lab, the top-2 margin of
// the float32 logits in x.d_crop_margin (pitch x1 - x0); logits in x.d_crop_logits (or d_logits)
static int referee_crop(Engine& e, ExactState& x, const uint8_t* d_img, int W, int y0, int x0, int y1, int x1, hipStream_t st,
                        float* d_logits = nullptr) {
    const int h = y1 - y0, w = x1 - x0;
    PSEG_TRY(xensure((void**)&x.d_crop_img, &x.crop_img_bytes, (size_t)h * w * e.in_ch));
    PSEG_TRY(xensure((void**)&x.d_crop_lab, &x.crop_lab_bytes, (size_t)h * w));
    PSEG_TRY(xensure((void**)&x.d_crop_margin, &x.crop_margin_bytes, (size_t)h * w * 4));
    if (!d_logits) {
        PSEG_TRY(xensure((void**)&x.d_crop_logits, &x.crop_logits_bytes, (size_t)h * w * e.n_classes * 4));
        d_logits = x.d_crop_logits;
    }
    PSEG_HIP(hipMemcpy2DAsync(x.d_crop_img, (size_t)w * e.in_ch, d_img + ((size_t)y0 * W + x0) * e.in_ch, (size_t)W * e.in_ch,
                              (size_t)w * e.in_ch, h, hipMemcpyDeviceToDevice, st));
    KnobScope ks(x.f32->e);
    return predict_device(x.f32->e, x.d_crop_img, h, w, d_logits, nullptr, nullptr, x.d_crop_lab, st, x.d_crop_margin);
}

static int calibrate(Engine& e, ExactState& x, const uint8_t* d_img, int H, int W, hipStream_t st) {
    // three crops (quarter points and centre of the page), 32-aligned origins, at most 384 x 384 each; the worst
    // logit difference of the three starts the threshold
    const int h = std::min(H, 384), w = std::min(W, 384);
    const size_t n = (size_t)h * w * e.n_classes;
    PSEG_TRY(xensure((void**)&x.d_la, &x.la_bytes, n * 4));
    PSEG_TRY(xensure((void**)&x.d_lb, &x.lb_bytes, n * 4));
    PSEG_HIP(hipMemsetAsync(x.d_counters, 0, 32, st));
    int done_y = -1, done_x = -1;
    for (int k = 1; k <= 3; ++k) {
        const int y0 = std::max(0, std::min(H - h, (H * k / 4 - h / 2))) & ~31, x0 = std::max(0, std::min(W - w, (W * k / 4 - w / 2))) & ~31;
        if (y0 == done_y && x0 == done_x) continue;     // small pages: the three crops coincide
        done_y = y0; done_x = x0;
        PSEG_TRY(referee_crop(e, x, d_img, W, y0, x0, y0 + h, x0 + w, st, x.d_la));
        PSEG_TRY(predict_device(e, x.d_crop_img, h, w, x.d_lb, nullptr, nullptr, nullptr, st, nullptr));   // the same crop as a page of its own
        max_abs_diff_kernel<<<(int)std::min<size_t>((n + 255) / 256, 2048), 256, 0, st>>>(x.d_la, x.d_lb, n, x.d_counters);
    }
    unsigned c[8];
    PSEG_HIP(hipMemcpyAsync(c, x.d_counters, 32, hipMemcpyDeviceToHost, st));
    PSEG_HIP(hipStreamSynchronize(st));
    memcpy(&x.calib_err, &c[2], 4);
    x.tau = std::max(std::max(4.0f * x.calib_err, 2.0f * x.margin_err), 1e-6f);
    if (const char* ev = PSEG_KNOB("PSEG_EXACT_TAU")) x.tau = (float)atof(ev);
    x.calib_Hp = round_up(H, 32);
    x.calib_Wp = round_up(W, 32);
    return PSEG_OK;
}

struct Rect { int by0, bx0, by1, bx1; };   // block coordinates, half open

// Cost model of the referee, in "float32 pixels": a crop costs its area (with halos, clipped to the page) but never less than
// what fills the chip once (a 16-launch float32 pass over a small crop is launch- and occupancy-bound), plus a fixed
// per-crop charge (crop copy, launches, the merge); the whole page costs its area plus the same charge.
constexpr double X_MIN_AREA = 96.0 * 1024, X_FIXED = 48.0 * 1024;
static double crop_cost(const Rect& r, int H, int W, int halo) {
    const int y0 = std::max(r.by0 * XB - halo, 0), x0 = std::max(r.bx0 * XB - halo, 0);
    const int y1 = std::min(r.by1 * XB + halo, H), x1 = std::min(r.bx1 * XB + halo, W);
    return std::max((double)(y1 - y0) * (x1 - x0), X_MIN_AREA) + X_FIXED;
}

// Rectangles over the to-do blocks: maximal horizontal runs per block row, runs of equal extent in consecutive rows
// stacked (a border ring becomes four strips, not its bounding box), then pairs merge -- first improvement, repeated
// until none is left -- while one crop over their bounding box is cheaper than the two (blocks whose halos overlap
// end up in one crop, tiny far-apart crops stay apart).  *cost = the model's price of the result; *union_px = the page
// area the halo-dilated blocks cover, a lower bound on what any cover must evaluate.
static std::vector<Rect> cover(const std::vector<uint8_t>& todo, int nby, int nbx, int H, int W, int halo, double* cost, double* union_px) {
    const int r = cdiv(halo, XB);
    std::vector<uint8_t> dil((size_t)nby * nbx, 0);
    for (int by = 0; by < nby; ++by)
        for (int bx = 0; bx < nbx; ++bx)
            if (todo[by * nbx + bx])
                for (int y = std::max(by - r, 0); y <= std::min(by + r, nby - 1); ++y)
                    for (int xx = std::max(bx - r, 0); xx <= std::min(bx + r, nbx - 1); ++xx) dil[y * nbx + xx] = 1;
    double up = 0;
    for (int by = 0; by < nby; ++by)
        for (int bx = 0; bx < nbx; ++bx)
            if (dil[by * nbx + bx]) up += (double)(std::min((by + 1) * XB, H) - by * XB) * (std::min((bx + 1) * XB, W) - bx * XB);
    *union_px = up;
    std::vector<Rect> rects, open;
    if (up >= 0.9 * (double)H * W) {    // nothing to gain from parts: one rectangle over everything
        rects.push_back(Rect{0, 0, nby, nbx});
        *cost = crop_cost(rects[0], H, W, halo);
        return rects;
    }
    for (int by = 0; by <= nby; ++by) {
        std::vector<Rect> next;
        if (by < nby)
            for (int bx = 0; bx < nbx;) {
                if (!todo[by * nbx + bx]) { ++bx; continue; }
                int b1 = bx;
                while (b1 < nbx && todo[by * nbx + b1]) ++b1;
                Rect run{by, bx, by + 1, b1};
                for (auto& o : open)
                    if (o.by1 == by && o.bx0 == bx && o.bx1 == b1) { run.by0 = o.by0; o.by1 = -1; break; }
                next.push_back(run);
                bx = b1;
            }
        for (auto& o : open) if (o.by1 >= 0) rects.push_back(o);
        open.swap(next);
    }
    for (bool again = rects.size() > 1; again;) {
        again = false;
        for (size_t i = 0; i < rects.size() && !again; ++i)
            for (size_t j = i + 1; j < rects.size(); ++j) {
                const Rect m{std::min(rects[i].by0, rects[j].by0), std::min(rects[i].bx0, rects[j].bx0), std::max(rects[i].by1, rects[j].by1),
                             std::max(rects[i].bx1, rects[j].bx1)};
                if (crop_cost(m, H, W, halo) < crop_cost(rects[i], H, W, halo) + crop_cost(rects[j], H, W, halo)) {
                    rects[i] = m;
                    rects.erase(rects.begin() + (ptrdiff_t)j);
                    again = true;
                    break;
                }
            }
    }
    double c = 0;
    for (auto& q : rects) c += crop_cost(q, H, W, halo);
    *cost = c;
    return rects;
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
    if (!x.d_counters) PSEG_HIP(hipMalloc((void**)&x.d_counters, 32));
    PSEG_TRY(sync_weights(e, x));
    const size_t npx = (size_t)H * W;
    float* d_margin = d_margin_out;
    if (!d_margin) { PSEG_TRY(xensure((void**)&x.d_margin, &x.margin_bytes, npx * 4)); d_margin = x.d_margin; }
    uint8_t* lab = d_labels_u8;
    x.st_flag_px = x.st_blocks = x.st_area = x.st_escal = x.st_full = x.st_rects = x.st_changed = x.st_cost = x.st_direct = 0;
    // A stream of pages that all end in the whole-page referee does not need the bf16 pass in front of it -- and a TEXT page always
    // ends there, whatever the first pass's precision: near-ties sit on class boundaries, the margin passes through zero on every one
    // of them, and a text page has one in three quarters of its 32-px blocks (profiles/r04_label_exact_study.json: 623 blocks hold a
    // pixel whose float32 margin is under 0.005, a hundredth of the bf16 pass's logit error).  After three such pages in a row the
    // next eight go to the float32 engine directly, then one page probes; every probe that falls back again doubles the stretch (up
    // to 512 pages), a page that stays partial ends it.  Not when the caller wants the margin map.
    if (x.skip_left > 0 && !d_margin_out && !PSEG_KNOB("PSEG_EXACT_TAU")) {
        --x.skip_left;
        KnobScope ks(x.f32->e);
        PSEG_TRY(predict_device(x.f32->e, d_img, H, W, nullptr, nullptr, nullptr, lab, st, nullptr));
        x.st_full = 1; x.st_area = 1.0; x.st_blocks = 1.0; x.st_cost = 1.0; x.st_direct = 1;
        if (d_labels) widen_u8_kernel<<<(int)std::min<size_t>((npx + 255) / 256, 8192), 256, 0, st>>>(lab, d_labels, npx);
        PSEG_HIP(hipGetLastError());
        return PSEG_OK;
    }
    // calibration: after a weight change, and again when the canvas has changed (another page format may carry other
    // content) -- at most every eighth page of a mixed-size stream: the running margin error from the refereed crops of
    // EVERY page is what tracks the stream, the calibration crops only seed it
    ++x.pages_since_calib;
    if (x.tau <= 0.0f || ((x.calib_Hp != round_up(H, 32) || x.calib_Wp != round_up(W, 32)) && x.pages_since_calib >= 8)) {
        x.pages_since_calib = 0;
        const float keep = x.tau;
        PSEG_TRY(calibrate(e, x, d_img, H, W, st));
        if (!PSEG_KNOB("PSEG_EXACT_TAU")) x.tau = std::max(x.tau, keep);     // a threshold the referee has raised stays raised until the weights change
    }
    // 1. throughput pass with the margin map
    PSEG_TRY(predict_device(e, d_img, H, W, nullptr, nullptr, nullptr, lab, st, d_margin));
    const int nby = cdiv(H, XB), nbx = cdiv(W, XB), nblk = nby * nbx;
    PSEG_TRY(xensure((void**)&x.d_flags, &x.flags_bytes, (size_t)nblk));
    PSEG_TRY(xensure((void**)&x.d_blockmin, &x.blockmin_bytes, (size_t)nblk * 4));
    x.h_blockmin.assign(nblk, 0.0f);
    x.h_flags.assign(nblk, 0);
    x.h_done.assign(nblk, 0);
    bool full = false;
    double area = 0, spent = 0;
    const int XHALO = halo_of(e);
    const double full_cost = (double)npx + X_FIXED;
    for (int iter = 0; iter < 6 && !full; ++iter) {
        PSEG_HIP(hipMemsetAsync(x.d_counters, 0, 32, st));
        flag_blocks_kernel<<<dim3(nbx, nby), 256, 0, st>>>(d_margin, H, W, x.tau, x.d_flags, x.d_blockmin, nbx, x.d_counters);
        PSEG_HIP(hipMemcpyAsync(x.h_flags.data(), x.d_flags, nblk, hipMemcpyDeviceToHost, st));
        PSEG_HIP(hipMemcpyAsync(x.h_blockmin.data(), x.d_blockmin, (size_t)nblk * 4, hipMemcpyDeviceToHost, st));
        unsigned c[8];
        PSEG_HIP(hipMemcpyAsync(c, x.d_counters, 32, hipMemcpyDeviceToHost, st));
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
        double cost = 0, union_px = 0;
        const std::vector<Rect> rects = cover(todo, nby, nbx, H, W, XHALO, &cost, &union_px);
        x.st_cost = (spent + cost) / full_cost;
        // the whole page is cheaper than the crops (class boundaries through most blocks, or tau escalated that far)
        if (spent + cost >= full_cost || PSEG_KNOB("PSEG_EXACT_FULL")) { full = true; break; }
        spent += cost;
        for (auto& r : rects) {
            const int y0 = std::max(r.by0 * XB - XHALO, 0), x0 = std::max(r.bx0 * XB - XHALO, 0);
            const int y1 = std::min(r.by1 * XB + XHALO, H), x1 = std::min(r.bx1 * XB + XHALO, W);
            area += (double)(y1 - y0) * (x1 - x0);
            PSEG_TRY(referee_crop(e, x, d_img, W, y0, x0, y1, x1, st));
            const int oy0 = r.by0 * XB, ox0 = r.bx0 * XB, oh = std::min(r.by1 * XB, H) - oy0, ow = std::min(r.bx1 * XB, W) - ox0;
            referee_merge_kernel<<<cdiv(oh * ow, 256), 256, 0, st>>>(x.d_crop_lab, x.d_crop_margin, x1 - x0, y0, x0, oy0, ox0, oh, ow, lab, W,
                                                                      d_margin, x.tau, x.d_counters);
            for (int by = r.by0; by < r.by1; ++by)
                for (int bx = r.bx0; bx < r.bx1; ++bx) x.h_done[by * nbx + bx] = 1;    // the whole rectangle now carries float32 labels
        }
        x.st_rects += (double)rects.size();
        PSEG_HIP(hipMemcpyAsync(c, x.d_counters, 32, hipMemcpyDeviceToHost, st));
        PSEG_HIP(hipStreamSynchronize(st));
        x.st_changed += c[3];
        float dm;                         // largest change of a margin between the two engines over the pixels just refereed
        memcpy(&dm, &c[4], 4);
        x.margin_err = std::max(x.margin_err, dm);
        float want = std::max(x.tau, 2.0f * x.margin_err);
        if (c[0] != 0) {                  // an "unflagged" pixel flipped: the threshold was too small, also for later pages
            float worst;
            memcpy(&worst, &c[2], 4);
            want = std::max(want, std::max(2.0f * x.tau, 2.0f * worst));
        }
        if (!(want > x.tau)) break;       // the threshold held against everything the referee saw
        x.tau = want;
        x.st_escal += 1;
        if (iter == 5) full = true;
    }
    int ndone = 0;
    for (int i = 0; i < nblk; ++i) ndone += x.h_done[i];
    x.st_blocks = (double)ndone / nblk;
    x.st_area = area / (double)npx;
    if (full) {
        // near-ties in most blocks (untrained weights, text pages at their line pitch): the referee takes the whole page
        KnobScope ks(x.f32->e);
        PSEG_TRY(predict_device(x.f32->e, d_img, H, W, nullptr, nullptr, nullptr, lab, st, nullptr));
        x.st_full = 1;
        x.st_area = 1.0;
        x.st_blocks = 1.0;
        if (x.skip_len > 0) { x.skip_len = std::min(2 * x.skip_len, 512); x.skip_left = x.skip_len; }     // a probe behind a direct stretch: back off
        else if (++x.fallback_streak >= 3) { x.skip_len = x.skip_left = 8; x.fallback_streak = 0; }
    } else {
        x.fallback_streak = 0;
        x.skip_len = 0;
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
    if (!e.exact) e.exact = new ExactState();
    ExactState& x = *(ExactState*)e.exact;
    // one persistent device slab (page | uint8 labels | int64 labels), grown on demand, freed with the engine
    const size_t npx = (size_t)H * W;
    const size_t off64 = (npx * e.in_ch + npx + 7) & ~(size_t)7;
    PSEG_TRY(xensure((void**)&x.d_page, &x.page_bytes, off64 + (labels ? npx * 8 : 0)));
    uint8_t* d_img = x.d_page;
    uint8_t* d_u8 = d_img + npx * e.in_ch;
    int64_t* d_i64 = labels ? (int64_t*)(d_img + off64) : nullptr;
    PSEG_HIP(hipMemcpyAsync(d_img, img, npx * e.in_ch, hipMemcpyHostToDevice, e.stream));
    PSEG_TRY(exact_labels(e, d_img, H, W, d_u8, d_i64, nullptr, e.stream));
    if (labels_u8) PSEG_HIP(hipMemcpyAsync(labels_u8, d_u8, npx, hipMemcpyDeviceToHost, e.stream));
    if (labels) PSEG_HIP(hipMemcpyAsync(labels, d_i64, npx * 8, hipMemcpyDeviceToHost, e.stream));
    return engine_status(e, e.stream);
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

int pseg_label_exact_stats_ex(const pseg_engine* h, double* out, int cap) {
    if (!h || !out || cap < 0) return fail(PSEG_EINVAL, "bad argument");
    double v[13] = {0};
    PSEG_TRY(pseg_label_exact_stats(h, v));
    if (const auto* x = (const ExactState*)h->e.exact) { v[8] = x->margin_err; v[9] = x->st_rects; v[10] = x->st_cost; v[11] = XB; v[12] = x->st_direct; }
    for (int i = 0; i < cap; ++i) out[i] = i < 13 ? v[i] : 0.0;
    return PSEG_OK;
}

}  // extern "C"
