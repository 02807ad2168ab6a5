// pseg_chain.hip -- the Predictor's chain as one device-resident call (lib/predictor.py:32-54):
//   Network.predict_single_data (argmax labels)  ->  [scale_to_original_shape: nearest resize of the label map,
//   lib/output.py:63-79]  ->  the post-processors in PredictSettings.post_process order (lib/postprocess.py:9-42)  ->
//   [generate_output_masks, lib/output.py:44-60].
// The reference hands a NumPy int64 map from stage to stage; here the uint8 label map stays in HBM between the stages
// (rounds 1-2 sent it down and up again around every stage: 25 MB each way per stage at 2048x1536), the page and the
// binarisation go up once, and only what the caller asked for comes down -- by DMA straight into the caller's arrays when
// those are page-locked (pseg_host_alloc / the Python shim's pooled pinned arrays).
#include <algorithm>

#include "pseg_common.h"

namespace pseg {

struct ChainState {
    hipStream_t s_aux = nullptr;        // uploads of the binarisation / colour table beside the network's kernels
    hipEvent_t ev_aux = nullptr;
    uint8_t* d_buf[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t cap[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};
enum { CB_IMG = 0, CB_LAB = 1, CB_LAB2 = 2, CB_BIN = 3, CB_MASKS = 4, CB_LUT = 5, CB_I64 = 6 };

static int censure(ChainState& c, int slot, size_t bytes) {
    if (c.cap[slot] >= bytes && c.d_buf[slot]) return PSEG_OK;
    if (c.d_buf[slot]) (void)hipFree(c.d_buf[slot]);
    c.d_buf[slot] = nullptr;
    c.cap[slot] = 0;
    PSEG_HIP(hipMalloc((void**)&c.d_buf[slot], bytes));
    c.cap[slot] = bytes;
    return PSEG_OK;
}

void chain_free(Engine& e) {
    auto* c = (ChainState*)e.chain;
    if (!c) return;
    for (int i = 0; i < 8; ++i) if (c->d_buf[i]) (void)hipFree(c->d_buf[i]);
    if (c->ev_aux) (void)hipEventDestroy(c->ev_aux);
    if (c->s_aux) (void)hipStreamDestroy(c->s_aux);
    delete c;
    e.chain = nullptr;
}

__global__ void chain_widen_kernel(const uint8_t* in, int64_t* out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}

}  // namespace pseg

using namespace pseg;

extern "C" int pseg_predict_chain(pseg_engine* h, const uint8_t* img, int H, int W, int Ho, int Wo, const uint8_t* binary,
                                  const int* post_ops, int n_post, unsigned flags, int64_t* labels, uint8_t* labels_u8,
                                  const uint8_t* lut, int n_lut, uint8_t* color, uint8_t* overlay, uint8_t* inverted,
                                  uint8_t* fg_color) {
    if (!h || !img) return fail(PSEG_EINVAL, "NULL argument");
    KnobScope knob_scope(h->e);
    Engine& e = h->e;
    if (H <= 0 || W <= 0 || n_post < 0 || (n_post > 0 && !post_ops)) return fail(PSEG_EINVAL, "bad argument");
    if (e.n_classes > 256) return fail(PSEG_EUNSUPPORTED, "the chain keeps a uint8 label map (<= 256 classes)");
    if (flags & ~(unsigned)PSEG_CHAIN_EXACT_LABELS) return fail(PSEG_EINVAL, "unknown flag bits 0x%x", flags);
    const bool resize = Ho > 0 && Wo > 0 && (Ho != H || Wo != W);
    const int Hl = resize ? Ho : H, Wl = resize ? Wo : W;
    const bool want_masks = color || overlay || inverted || fg_color;
    bool need_bin = want_masks;
    for (int i = 0; i < n_post; ++i) {
        if (post_ops[i] != PSEG_POST_CC_VOTE && post_ops[i] != PSEG_POST_BBOX) return fail(PSEG_EINVAL, "unknown post-processor id %d", post_ops[i]);
        need_bin |= post_ops[i] == PSEG_POST_CC_VOTE;
    }
    if (need_bin && !binary) return fail(PSEG_EINVAL, "the vote / the masks need the binarisation");
    if (want_masks && (!lut || n_lut < 1)) return fail(PSEG_EINVAL, "the masks need the colour table");
    PSEG_HIP(hipSetDevice(e.device));
    if (!e.chain) {
        auto* nc = new ChainState();
        e.chain = nc;
        PSEG_HIP(hipStreamCreateWithFlags(&nc->s_aux, hipStreamNonBlocking));
        PSEG_HIP(hipEventCreateWithFlags(&nc->ev_aux, hipEventDisableTiming));
    }
    ChainState& c = *(ChainState*)e.chain;
    hipStream_t st = e.stream;
    const size_t npx = (size_t)H * W, nl = (size_t)Hl * Wl;
    const size_t nla = (nl + 255) & ~(size_t)255;          // buffer stride: the mask / vote kernels want 4-byte aligned maps
    // a reallocation must not race with the previous call's work: every call ends synchronised, so the buffers are idle here
    PSEG_TRY(censure(c, CB_IMG, npx * e.in_ch));
    PSEG_TRY(censure(c, CB_LAB, npx));
    PSEG_TRY(censure(c, CB_LAB2, 2 * nla));          // resize target + bounding-box ping-pong
    if (need_bin) PSEG_TRY(censure(c, CB_BIN, nl));
    if (want_masks) { PSEG_TRY(censure(c, CB_MASKS, 4 * nla * 3)); PSEG_TRY(censure(c, CB_LUT, (size_t)n_lut * 3)); }
    if (labels) PSEG_TRY(censure(c, CB_I64, nl * 8));
    // every way out -- also an error return in the middle -- ends with both streams drained: copies from / to the caller's host
    // arrays must not be in flight when the caller gets its buffers back
    struct Drain { hipStream_t a, b; ~Drain() { (void)hipStreamSynchronize(a); (void)hipStreamSynchronize(b); } } drain{st, c.s_aux};
    // uploads: the page on the engine's stream (the network waits for it anyway), binarisation and colour table beside it
    PSEG_HIP(hipMemcpyAsync(c.d_buf[CB_IMG], img, npx * e.in_ch, hipMemcpyHostToDevice, st));
    if (need_bin) PSEG_HIP(hipMemcpyAsync(c.d_buf[CB_BIN], binary, nl, hipMemcpyHostToDevice, c.s_aux));
    if (want_masks) PSEG_HIP(hipMemcpyAsync(c.d_buf[CB_LUT], lut, (size_t)n_lut * 3, hipMemcpyHostToDevice, c.s_aux));
    if (need_bin) PSEG_HIP(hipEventRecord(c.ev_aux, c.s_aux));
    // 1. the network: uint8 argmax labels (float32 engine: bit-exact; bf16 engine: throughput labels, or the label-exact mode)
    if ((flags & PSEG_CHAIN_EXACT_LABELS) && e.mode == PSEG_MODE_BF16)
        PSEG_TRY(pseg_predict_exact_labels_device(h, c.d_buf[CB_IMG], H, W, c.d_buf[CB_LAB], nullptr, nullptr, st));
    else
        PSEG_TRY(predict_device(e, c.d_buf[CB_IMG], H, W, nullptr, nullptr, nullptr, c.d_buf[CB_LAB], st, nullptr));
    uint8_t* cur = c.d_buf[CB_LAB];
    uint8_t* const bufA = c.d_buf[CB_LAB2];
    uint8_t* const bufB = c.d_buf[CB_LAB2] + nla;
    // 2. scale_to_original_shape: order-0 gather of the label map (preserving_resize(pred, original_shape))
    if (resize) {
        PSEG_TRY(pseg_resize_nearest_device(e.device, cur, H, W, 1, bufA, Hl, Wl, st));
        cur = bufA;
    }
    // 3. post-processors, in order
    if (need_bin) PSEG_HIP(hipStreamWaitEvent(st, c.ev_aux, 0));
    for (int i = 0; i < n_post; ++i) {
        if (post_ops[i] == PSEG_POST_CC_VOTE) {
            PSEG_TRY(pseg_cc_vote_device_u8(e.device, cur, c.d_buf[CB_BIN], Hl, Wl, e.n_classes, st));
        } else {
            uint8_t* const dst = cur == bufA ? bufB : bufA;
            PSEG_TRY(pseg_bbox_fill_device_u8(e.device, cur, dst, Hl, Wl, e.n_classes, st));
            cur = dst;
        }
    }
    // 4. outputs
    if (labels_u8) PSEG_HIP(hipMemcpyAsync(labels_u8, cur, nl, hipMemcpyDeviceToHost, st));
    if (labels) {
        chain_widen_kernel<<<(int)std::min<size_t>((nl + 255) / 256, 8192), 256, 0, st>>>(cur, (int64_t*)c.d_buf[CB_I64], nl);
        PSEG_HIP(hipMemcpyAsync(labels, c.d_buf[CB_I64], nl * 8, hipMemcpyDeviceToHost, st));
    }
    if (want_masks) {
        uint8_t* m = c.d_buf[CB_MASKS];
        uint8_t* dm[4] = {color ? m : nullptr, overlay ? m + nla * 3 : nullptr, inverted ? m + 2 * nla * 3 : nullptr, fg_color ? m + 3 * nla * 3 : nullptr};
        PSEG_TRY(pseg_masks_device_u8(e.device, cur, c.d_buf[CB_BIN], c.d_buf[CB_LUT], n_lut, Hl, Wl, dm[0], dm[1], dm[2], dm[3], st));
        uint8_t* hm[4] = {color, overlay, inverted, fg_color};
        for (int k = 0; k < 4; ++k)
            if (hm[k]) PSEG_HIP(hipMemcpyAsync(hm[k], dm[k], nl * 3, hipMemcpyDeviceToHost, st));
    }
    PSEG_HIP(hipStreamSynchronize(c.s_aux));
    return engine_status(e, st);
}
