// pseg_common.h -- shared declarations of the libpseg.so sources (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#include "../../include/pseg.h"

struct pseg_engine;
namespace pseg {

// ---- error plumbing (thread-local message, int status: SURVEY.md 8b "Errors") -------------
std::string& last_error();
int fail(int code, const char* fmt, ...);

#define PSEG_HIP(expr)                                                                         \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            return ::pseg::fail(PSEG_EHIP, "%s failed: %s (%s:%d)", #expr,                     \
                                hipGetErrorString(_e), __FILE__, __LINE__);                    \
    } while (0)

#define PSEG_TRY(expr)                                                                         \
    do {                                                                                       \
        int _rc = (expr);                                                                      \
        if (_rc != PSEG_OK) return _rc;                                                        \
    } while (0)

// Knobs.  Two kinds, one lookup (PSEG_KNOB):
//   * ENVIRONMENT knobs -- the PSEG_ENV_KNOBS list below, fourteen names, documented in README.md: operational choices a deployment
//     may make (page-unit size, the persistent kernel families off, checks, logging).  Nothing else in the process environment
//     reaches the release library.
//   * PLAN SWITCHES -- alternative kernel / fusion choices that must not change results beyond the documented bars; every one is
//     exercised by a bit-identity or tolerance test.  They exist for those tests and for A/B measurements and enter ONLY through
//     pseg_create_plan's `switches` string ("PSEG_NO_DQ=1;PSEG_WS_FORM=2"); the Python test harness builds that string from
//     os.environ when pseg_amd.engine.PLAN_FROM_ENV is set (tests/conftest.py, tools/), a product caller never does.
// pseg_create* takes ONE snapshot (the listed environment knobs + the plan switches) for the new engine; the engine keeps it for
// its whole life (std::shared_ptr) and every entry point that takes an engine answers queries from the engine's OWN snapshot
// (KnobScope, thread-local) -- plan-time and launch-time reads of one engine always agree, and neither a later change of the
// environment nor the creation of another engine (the float32 companion of the label-exact mode inherits its parent's snapshot)
// can alter a running engine.  Engine-less entries (post-process, resize) read the newest environment snapshot.  Each call site
// caches its answer per snapshot id.
// Knobs that produce WRONG results (timing ablations: PSEG_DBG, PSEG_XM_DBG, PSEG_SP_DBG, PSEG_PP_NODMA, PSEG_PP_NOEPI, the
// in-kernel trace stamps) exist only in the diagnostic build (libpseg_diag.so, -DPSEG_DIAG=1, which reads everything from the
// environment): PSEG_DIAG_KNOB is a constant nullptr in the release library (tests/test_abi.py checks that the release .so does
// not even hold the names).
#define PSEG_ENV_KNOBS                                                                                                          \
    "PSEG_BATCH_PAGES", "PSEG_NO_PAGE_BATCH", "PSEG_GENERIC", "PSEG_NO_SP", "PSEG_NO_PP", "PSEG_NO_WS", "PSEG_SP_CHECK",        \
    "PSEG_EXACT_TAU", "PSEG_EXACT_FULL", "PSEG_TRAIN_ONE_STREAM", "PSEG_TRAIN_STRICT", "PSEG_LOG_GENERIC", "PSEG_LOG_SP",       \
    "PSEG_CCL_GLOBAL"
#ifndef PSEG_DIAG
#define PSEG_DIAG 0
#endif
struct KnobSnap;                                  // id + the PSEG_* variables at the time of the snapshot
std::shared_ptr<const KnobSnap> knobs_snapshot(const char* plan = nullptr); // the listed environment knobs now (+ plan switches "A=1;B=2"); without a plan it also becomes the "newest" one
unsigned knob_generation();                       // id of the snapshot in scope (the engine's, else the newest), >= 1
const char* knob_lookup(const char* name);        // value in that snapshot, or nullptr
struct Engine;
struct KnobScope {                                // RAII: knob queries on this thread read `e`'s snapshot
    explicit KnobScope(const Engine& e);
    ~KnobScope();
    const KnobSnap* prev;
};
#if PSEG_DIAG
#define PSEG_KNOB(name) getenv(name)
#define PSEG_DIAG_KNOB(name) getenv(name)
#else
#define PSEG_KNOB(name)                                                                                   \
    ([]() -> const char* {                                                                                \
        static thread_local unsigned gen_ = 0;       /* per thread: no torn (id, value) pairs */         \
        static thread_local const char* v_ = nullptr;                                                     \
        const unsigned g_ = ::pseg::knob_generation();                                                    \
        if (gen_ != g_) {                                                                                 \
            v_ = ::pseg::knob_lookup(name);                                                               \
            gen_ = g_;                                                                                    \
        }                                                                                                 \
        return v_;                                                                                        \
    }())
#define PSEG_DIAG_KNOB(name) ((const char*)nullptr)
#endif

constexpr int PSEG_CHAIN_BLOCK = 16;   // float32 mode: input channels per pass of the accumulation chain (oracle/pseg_oracle.c ORC_CHAIN_BLOCK)
constexpr int PSEG_MAXC = 64;   // classes the train-step metric slots and the wide bf16 logits kernel are sized for
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return cdiv(a, b) * b; }

// ---- graph description ---------------------------------------------------------------------
enum OpType { OP_CONV = 0, OP_DECONV2 = 1, OP_POOL = 2, OP_LOGITS = 3, OP_BN = 4 };

struct Tensor {
    std::string name;   // producing Keras layer name ("input", "conv2d", "max_pooling2d", ...)
    int s = 0;          // log2 down-scale relative to the padded canvas
    int C = 0;          // true channel count
    int Cs = 0;         // storage channels: C (f32 mode) or round_up(C, 8) (bf16 mode)
    void* d = nullptr;  // device buffer, NHWC, (Hp>>s) x (Wp>>s) x Cs -- of the page slot in use (Engine::pages slots; slot 0 = base)
    void* base = nullptr;       // the allocation: `pages` slots of page_bytes each (bf16 page batches), d == base outside a per-page run
    size_t page_bytes = 0;      // bytes of one page slot at the current canvas
    size_t bytes = 0;           // bytes allocated
    bool fused = false; // bf16 mode: never written to HBM (lives only inside a fused kernel)
    bool relu_stored = false;   // bf16 mode: stored after the pre-activation ReLU of its readers (mfma_plan_graph)
};

struct Param {
    std::string name;   // "conv2d/kernel", ...
    int64_t shape[4] = {0, 0, 0, 0};
    int ndim = 0;
    std::vector<float> host;  // Keras layout
    bool set = false;
};

struct Op {
    int type = OP_CONV;
    std::string layer;   // Keras layer name
    int k = 1, stride = 1;
    bool transposed = false;  // Conv2DTranspose stride 1: flip + swap channels at upload
    int src0 = -1, src1 = -1; // concat [src0, src1]
    int up0 = 0, up1 = 0;     // nearest x2 upsample folded into the gather
    int in_relu = 0, relu = 0;
    int add = -1;             // residual addend tensor (Add()), applied after bias
    int dst = -1;
    int pool_dst = -1;        // bf16 mode: fused 2x2 max-pool output
    int relu_dst = -1;        // bf16 mode: second output tensor holding max(x, 0) for the pre-activation readers (mfma_plan_graph)
    int kparam = -1, bparam = -1;   // OP_BN: gamma / beta of the BatchNormalization layer
    int mmparam = -1, mvparam = -1; // OP_BN: moving_mean / moving_variance
    int bn_c0 = 0;                  // OP_BN: first channel of this op's slice of the layer's vectors (a BN over a
                                    // Concatenate runs as one op per source tensor)
    int Cin = 0, Cout = 0;
    // device weights
    float* d_w = nullptr;     // f32 correlation form [KH][KW][Cin][Cout] (+slack) / [2][2][Cin][Cout]; OP_BN: [gamma|beta|mean|var][C]
    float* d_b = nullptr;     // f32 bias; OP_BN: [batch mean | 1/sqrt(var+eps)][C] of the last training forward, then 4*C doubles of reduction scratch
    float* d_wrem = nullptr;  // f32 mode: the left-over output channels' kernel as shifted copies (conv_xb_kernel REM), owned by the op,
    size_t wrem_bytes = 0;    //   sized at upload_weights, rebuilt by the launcher whenever wrem_valid is false (weights changed)
    bool wrem_valid = false;
    void* plan = nullptr;     // bf16 mode: MfmaPlan (pseg_mfma.hip), owned by the op
    bool fused_away = false;  // bf16 mode: op folded into a neighbour (pool -> conv epilogue, logits -> deconv tail)
    int fuse1 = -1;           // bf16 mode: OP_CONV that recomputes this first-layer op on its halo tile
    int tail_logits = -1;     // bf16 mode: OP_DECONV2 that also runs this OP_LOGITS (fused tail)
    int dq_fuse = -1;         // bf16 mode: OP_CONV (k5, 80 couts) whose epilogue also runs this later OP_DECONV2 (k2 s2) on its accumulators; the deconv is fused_away
    int into_tail = -1;       // bf16 mode: OP_DECONV2 (ReLU) computed inside the composed-tail kernel of this later OP_DECONV2 (fused_away)
    bool pool_only = false;   // bf16 mode: the full-resolution output is read by nothing but the fused pool: it is not stored
    int skiplog = -1;         // bf16 mode: conv whose output only feeds (a fused pool and) this OP_LOGITS as the skip: it stores
                              // its logits contribution (f32, padded classes) instead of the full-resolution tensor
    int nw_hint = 4;          // bf16 mode: waves per workgroup the plan should be packed for (4 or 8)
    float dropout = 0.0f;     // Dropout(rate) on this op's output: identity at inference, applied by the train step
    double flops_per_canvas_px = 0;  // algorithmic, true channels
    int timing_slot = -1;
};

// ---- bf16 MFMA path (pseg_mfma.hip) ----------------------------------------------------------
struct Engine;
// host-side packing + upload; w = correlation-form f32 weights as uploaded for the exact path
int mfma_pack_op(Engine& e, Op& op, const std::vector<float>& w, const std::vector<float>& bias);
void mfma_free_op(Op& op);
void mfma_trim_op(Op& op);                                   // frees the plan's canvas-sized buffers (pseg_engine_trim)
int mfma_plan_graph(Engine& e);                              // pool fusion etc., once per engine
int mfma_launch_conv(Engine& e, Op& op, hipStream_t st);     // OP_CONV
int mfma_launch_deconv2(Engine& e, Op& op, hipStream_t st);  // OP_DECONV2
int mfma_launch_pool(Engine& e, Op& op, hipStream_t st);     // OP_POOL (unfused fallback)
int mfma_launch_logits(Engine& e, Op& op, float* d_logits, float* d_probs, int64_t* d_labels,
                       uint8_t* d_labels_u8, hipStream_t st);
int mfma_preprocess(Engine& e, const uint8_t* d_img, hipStream_t st);

struct TimingSlot {
    std::string name;
    double flops = 0;   // algorithmic flops of one launch at the current canvas
    double total_ms = 0;
    int64_t launches = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

struct Engine {
    std::shared_ptr<const KnobSnap> knobs;   // the PSEG_* snapshot this engine was created under (see PSEG_KNOB)
    int arch = 0, n_classes = 0, in_ch = 1, device = 0, mode = 0;
    unsigned flags = 0;            // PSEG_FLAG_* of pseg_create_ex
    bool bn_training = false;      // float32 engine, while a TRAINING forward runs: BatchNormalization layers use batch statistics and update their moving ones
    hipStream_t stream = nullptr;
    std::vector<Tensor> tensors;
    std::vector<Param> params;
    std::vector<Op> ops;
    int input_tensor = 0;
    // canvas
    int H = 0, W = 0, Hp = 0, Wp = 0;
    int pages = 1;                 // page slots every activation tensor has room for (bf16 page batches: pseg_predict_batch)
    int page = 0;                  // page slot the per-page launches of a batch run are working on
    int batch_pages = 0;           // > 1 while a launch covers that many page slots at once (a tile index carries the page)
    bool weights_dirty = true;
    bool exact_dirty = true;       // label-exact mode: the float32 companion's weights / the calibrated threshold are stale
    float* d_lut = nullptr;        // 256-entry u/255 table (f32)
    const uint8_t* cur_img = nullptr;  // bf16 mode: the uint8 page of the running predict call
    float* cur_logits = nullptr;       // bf16 mode: output pointers of the running predict call
    float* cur_probs = nullptr;
    int64_t* cur_labels = nullptr;
    uint8_t* cur_labels_u8 = nullptr;
    float* cur_margin = nullptr;       // bf16 mode: top-1 minus top-2 logit map requested by the running call (label-exact mode)
    bool margin_done = false;          // ... and whether the tail kernel of the graph wrote it (else it is derived from the logits)
    float* d_logits_tmp = nullptr; // H*W*C f32 when the caller does not want logits
    size_t logits_tmp_bytes = 0;
    uint8_t* d_img_stage = nullptr;
    size_t img_stage_bytes = 0;
    int64_t* d_lab_stage = nullptr;
    float* d_prob_stage = nullptr;
    float* d_logit_stage = nullptr;
    size_t lab_stage_bytes = 0, prob_stage_bytes = 0, logit_stage_bytes = 0;
    void* train = nullptr;   // TrainState (pseg_train.hip), f32 mode only
    void* exact = nullptr;   // ExactState (pseg_exactlabels.hip): float32 companion engine, margin / flag buffers of the label-exact mode
    void* batch = nullptr;   // BatchState (pseg_predict_batch): copy streams, events, two staging slots
    void* chain = nullptr;   // ChainState (pseg_predict_chain): device buffers of the Predictor chain
    void* dist = nullptr;    // DistState (pseg_allreduce_init): RCCL communicator of the data-parallel train step
    int relaxed_f32 = 0;     // != 0 during a train / eval step: wide float32 layers may run channel-blocked on the matrix cores
    uint32_t drop_key = 0;   // != 0 while a TRAINING forward runs: Dropout layers are live (key = seed / step mix)
    const float* cur_img_f32 = nullptr;   // float32 exact mode: float page (0..255 scale) instead of the uint8 one (augmented training samples)
    int* d_sp_err = nullptr;   // bf16 mode: the give-up record of conv_sp_kernel's bounded counter waits (8 ints: code, wave, need, have,
    int* h_sp_err = nullptr;   //   need2, have2, workgroup, op index), zero while all is well; h_: its pinned read-back slot (engine_status)
    // timing
    bool timing = false;
    std::vector<TimingSlot> slots;
    std::vector<hipEvent_t> event_pool;

    int tH(const Tensor& t) const { return Hp >> t.s; }
    int tW(const Tensor& t) const { return Wp >> t.s; }
};

// f32-exact conv launcher shared with the training path (pseg_train.hip)
struct ConvArgs {
    const float* src0;
    const float* src1;
    int C0, C1;
    int up0, up1;
    int Hin, Win;  // logical input dims (after the folded upsample)
    const float* w;
    const float* bias;
    const float* add;
    float* dst;
    int KH, KW, stride, pt, pl, Hout, Wout, Cout, in_relu, relu;
    const float* mask;   // training dgrad: input value counts only where mask (same layout as src0) > 0
    int dst_pitch;       // pixels per output row (0: Wout) -- crop / canvas-pitched outputs
    int out_sy, out_sx, out_oy, out_ox;   // MFMA kernel only: output pixel (y*sy + oy, x*sx + ox); 0 strides = 1
    int deconv4;         // MFMA kernel only: Conv2DTranspose k2 s2 as one GEMM, n = ab*Cout + co -> (2y + a, 2x + b)
    int dbg;             // MFMA kernel only: timing experiments (PSEG_XM_DBG: 1 no staging loads, 2 no stores, 4 no k-loop) -- wrong results
    float* pool_dst;     // blocked MFMA kernel, 8-row tiles only: also store the 2x2 max-pool of the output ((Hout/2) x (Wout/2) x Cout)
    int relaxed;         // MFMA kernel only: the caller accepts a channel-blocked summation order (train step) for layers whose all-channel tile does not fit LDS
    const float* wrem = nullptr;   // blocked MFMA kernel, set by its launcher: the left-over output channels' kernel, shifted copies [KH][KW + DX - 1][Cin][16] (conv_xb_kernel REM)
    float* wrem_buf = nullptr;     // caller-owned buffer for that kernel (wrem_bytes_for() bytes; null or too small: padded cout tiles instead)
    size_t wrem_cap = 0;
    bool* wrem_valid = nullptr;    // null: rebuild on every launch (scratch weights); else rebuilt when *wrem_valid is false, then set
};
// bytes of ConvArgs.wrem_buf a KH x KW, Cin -> Cout stride-1 layer needs for its left-over channel tile, 0 when it has none
size_t wrem_bytes_for(int KH, int KW, int Cin, int Cout);
int launch_conv_exact(const ConvArgs& a, hipStream_t st, bool* pooled = nullptr);   // *pooled: ConvArgs.pool_dst was written by the conv kernel
// HBM-bound float32 layers on the vector ALU (pseg_exact_valu.hip): first layer; Conv2DTranspose k2 s2, optionally with the
// logits layer + argmax behind it.  1 = launched, 0 = not a layer for these kernels, < 0 error.
struct TailArgs {
    const float* src0; const float* src1; int C0, C1, Hin, Win;
    const float* w; const float* bias; int Cout, relu;
    float* dst;
    const float* skip; int Cs;
    const float* wl; const float* bl; int ncls;
    int H, W;
    float* logits; int64_t* labels; uint8_t* labels_u8;
};
int launch_conv_first_valu(const ConvArgs& a, hipStream_t st);
int launch_deconv2_valu(const TailArgs& a, bool tail, hipStream_t st);
int launch_conv_exact_mfma(const ConvArgs& a, hipStream_t st);   // 1 launched, 0 does not fit, < 0 error
// split form of UpSampling2D(2) -> Conv2D(k2) (pseg_upsplit.hip)
struct UpSplit;
int upsplit_create(UpSplit** out, const std::vector<float>& w, const std::vector<float>& bias, int Cin, int Cs0, int Cout, int CoS);
void upsplit_free(UpSplit* u);
int upsplit_launch(UpSplit* u, const uint16_t* src, int Hs, int Ws, uint16_t* dst, int relu, hipStream_t st);
// BatchNormalization (pseg_bn.hip), NHWC float32 tensors of npx pixels x C channels; `par` = [gamma|beta|moving_mean|moving_var][C]
constexpr float PSEG_BN_EPS = 1e-3f;        // tf.keras.layers.BatchNormalization defaults (lib/model.py:268,315 pass none)
constexpr float PSEG_BN_MOMENTUM = 0.99f;
int bn_infer(const float* x, float* y, size_t npx, int C, const float* par, float* saved, int relu, hipStream_t st);
int bn_train_forward(const float* x, float* y, size_t npx, int C, float* par, float* saved, int relu, int up, hipStream_t st);
int bn_backward(const float* x, const float* y_mask, const float* dy, float* dx_accum, size_t npx, int C, const float* par,
                float* saved, float* dgamma, float* dbeta, hipStream_t st);
int bn_infer_bf16(const uint16_t* x, uint16_t* y, size_t npx, int Cs, const float* scale_shift, int relu, hipStream_t st);
size_t bn_saved_bytes(int C);
// Dropout mask of element i under `key`: keep iff hash(i, key) >= rate (inverted dropout, kept values * 1/(1-rate))
void launch_dropout(float* x, size_t n, uint32_t key, float rate, hipStream_t st);
int ccl_roots(const uint8_t* d_bin, int* d_L, int H, int W, int connectivity, hipStream_t st);   // pseg_post.hip
int upload_weights(Engine& e);
// one page through the engine's graph, device buffers, asynchronous on `st` (every output optional)
int predict_device(Engine& e, const uint8_t* d_img, int H, int W, float* d_logits, float* d_probs, int64_t* d_labels,
                   uint8_t* d_labels_u8, hipStream_t st, float* d_margin);
void launch_margin_from_logits(const float* d_logits, size_t n, int C, float* d_margin, hipStream_t st);
bool mfma_tail_emits_margin(const Engine& e);   // the bf16 graph's tail kernel writes the margin map itself
int create_engine(int arch, int n_classes, int in_channels, int device, int mode, unsigned flags,
                  std::shared_ptr<const KnobSnap> inherit, struct ::pseg_engine** out, const char* plan = nullptr);   // pseg_create_ex; `inherit` = a parent engine's knob snapshot
void exact_free(Engine& e);
void chain_free(Engine& e);
void dist_free(Engine& e);                      // RCCL communicator (pseg_dist.hip)                     // Predictor chain buffers (pseg_chain.hip)                     // label-exact mode state (pseg_exactlabels.hip)
int set_canvas(Engine& e, int H, int W, hipStream_t st, int pages = 1);
// Waits for `st`, then reports -- and clears -- the engine's device-side error record: PSEG_EHIP when a counter wait of
// conv_sp_kernel gave up since the last report (the label maps produced since then are not to be trusted).  Called by every
// entry that synchronises with the host anyway and by pseg_engine_status (for the callers of the asynchronous _device entries).
int engine_status(Engine& e, hipStream_t st);
bool mfma_op_batchable(const Engine& e, const Op& op);      // the op's kernel takes several page slots in one launch
int run_exact(Engine& e, const uint8_t* d_img, float* d_logits, float* d_probs, int64_t* d_labels,
              uint8_t* d_labels_u8, hipStream_t st);
// training state (pseg_train.hip)
int train_sync_weights_to_host(Engine& e);
void train_free(Engine& e);

int time_begin(Engine& e, Op& op, hipStream_t st, hipEvent_t* ev0);
int time_end(Engine& e, Op& op, hipStream_t st, hipEvent_t ev0);

}  // namespace pseg

// the opaque handle of include/pseg.h
struct pseg_engine {
    pseg::Engine e;
};
