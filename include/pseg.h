/*
 * pseg.h -- C ABI of libpseg.so, the MI355X (gfx950) engine for the per-pixel
 * page-segmentation hot path of ocr4all_pixel_classifier.
 *
 * The reference (pure Python over TensorFlow/OpenCV) has no FFI layer; the boundary this
 * library sits behind is the Python API of ocr4all_pixel_classifier.lib.
 * Each entry point names the reference interface it replaces.  Signatures carry plain
 * pointers and sizes only -- no torch / numpy types.  Every function returns 0 on success
 * and a negative PSEG_E* code on failure; pseg_last_error() returns a thread-local message
 * (the Python shim raises it as Exception, the reference's error style).
 *
 * Ownership: the library never retains a caller pointer past the call.  Device buffers are
 * owned by the opaque pseg_engine.  "_device" variants take pointers that are already
 * resident in HBM on the engine's device and enqueue on the given hipStream_t (passed as
 * void*; NULL = the engine's own stream) without synchronising.
 */
#ifndef PSEG_H
#define PSEG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSEG_ABI_VERSION 1

/* lib/architecture.py:6-11 -- the in-scope Architecture members. */
enum { PSEG_ARCH_FCN_SKIP = 0, PSEG_ARCH_FCN = 1, PSEG_ARCH_UNET = 2, PSEG_ARCH_RES_UNET = 3 };

/* Arithmetic mode.  F32_EXACT: float32 tensors, one sequential fmaf chain per output -- slabs of 16 input
 * channels, (ky,kx,ci) inside a slab -- bit-identical to oracle/pseg_oracle.c.  BF16: bf16 activations + bf16 kernels,
 * float32 MFMA accumulation (throughput mode, BASELINE.json configs[1]). */
enum { PSEG_MODE_F32_EXACT = 0, PSEG_MODE_BF16 = 1 };

enum {
    PSEG_OK = 0,
    PSEG_EINVAL = -1,   /* bad argument / shape */
    PSEG_ENOTFOUND = -2, /* unknown weight name */
    PSEG_EHIP = -3,     /* HIP runtime error (message has the hipError string) */
    PSEG_ENOMEM = -4,
    PSEG_EUNSUPPORTED = -5
};

typedef struct pseg_engine pseg_engine;

int pseg_abi_version(void);
const char* pseg_last_error(void);

/* Number of visible HIP devices (0 when none; never fails). */
int pseg_device_count(void);

/* ---- Network: lib/network.py:18-107 (graph construction + weight loading) ------------- */

/* Replaces model_constructor.model()([input, binary], n_classes) (lib/network.py:89) for
 * lib/model.py:45 (fcn_skip), :206 (fcn), :151 (unet), :237 (res_unet). */
int pseg_create(int arch, int n_classes, int in_channels, int device, int mode,
                pseg_engine** out);
/* ... with graph options.  PSEG_FLAG_BATCHNORM: tf.keras.layers.BatchNormalization (defaults: epsilon 1e-3, momentum
 * 0.99) at the sites the reference's constructors provide for it -- res_unet's bn_act (lib/model.py:265-271: in front
 * of every pre-activation ReLU and behind every shortcut convolution; the reference hard-wires its switch to False)
 * -- the same placement as conv_block_simple's Conv2D -> BatchNormalization -> ReLU (lib/model.py:310-317) behind the
 * shortcut convolutions.  Adds "batch_normalization[_N]/gamma|beta|moving_mean|moving_variance" to the weight table
 * (Keras order).  Prediction and pseg_eval_step use the moving statistics; pseg_train_forward_backward normalises with
 * the page's batch statistics, updates the moving ones and back-propagates through the layer.  Only
 * PSEG_ARCH_RES_UNET has such sites (PSEG_EUNSUPPORTED otherwise).  flags = 0 is pseg_create. */
enum { PSEG_FLAG_BATCHNORM = 1 };
int pseg_create_ex(int arch, int n_classes, int in_channels, int device, int mode, unsigned flags,
                   pseg_engine** out);
/* ... with PLAN SWITCHES: alternative kernel / fusion choices of the bf16 and float32 engines ("PSEG_NO_DQ=1;PSEG_WS_FORM=2";
 * NULL or "" = pseg_create_ex).  Test and measurement entry: every switch leaves the results within the documented bars (most
 * are bit-identical) and is pinned by a test; the process environment cannot set them -- the release library reads only the
 * names pseg_env_knobs() lists from the environment (README.md documents them), once per engine at creation. */
int pseg_create_plan(int arch, int n_classes, int in_channels, int device, int mode, unsigned flags, const char* switches,
                     pseg_engine** out);
/* the environment variables the release library reads, one per line (static storage) */
const char* pseg_env_knobs(void);
int pseg_destroy(pseg_engine* e);

/* Weight table in Keras creation order; names are Keras' default layer names plus
 * "/kernel" or "/bias" ("conv2d/kernel", "conv2d_transpose_1/bias", "logits/kernel", ...).
 * Layouts are Keras': Conv2D (kh,kw,Cin,Cout), Conv2DTranspose (kh,kw,Cout,Cin), bias (Cout). */
int pseg_num_weights(const pseg_engine* e);
int pseg_weight_info(const pseg_engine* e, int index, char* name, size_t name_cap,
                     int64_t shape[4], int* ndim);
/* Replaces model.load_weights (lib/network.py:106-107) / model.set_weights. */
int pseg_set_weights(pseg_engine* e, const char* name, const float* data, const int64_t* shape,
                     int ndim);
int pseg_get_weights(const pseg_engine* e, const char* name, float* out, int64_t count);

/* ---- Predict: lib/network.py:248-260 (Network.predict_single_data) -------------------- */

/* img: uint8 (H,W) network input (inverted, line-height normalised page).  Computes
 * x/255 -> pad to 32 -> FCN -> crop -> logits -> softmax / argmax.  Any of logits (H,W,C f32),
 * probs (H,W,C f32), labels (H,W int64) may be NULL.  Host pointers; synchronous. */
int pseg_predict(pseg_engine* e, const uint8_t* img, int H, int W, float* logits, float* probs,
                 int64_t* labels);

/* Same with device-resident buffers; asynchronous on `stream`.  labels_u8 is an optional
 * compact label map (n_classes <= 256). */
int pseg_predict_device(pseg_engine* e, const uint8_t* d_img, int H, int W, float* d_logits,
                        float* d_probs, int64_t* d_labels, uint8_t* d_labels_u8, void* stream);

/* Predictor.predict's page loop (lib/predictor.py:27-30) for n_pages pages of ONE shape that are resident on the device:
 * d_imgs = the pages one behind the other (n_pages * H * W * in_channels bytes), d_labels / d_labels_u8 = the label maps
 * one behind the other (either may be NULL).  bf16 engines keep a page slot per page in every activation tensor and give
 * the low-resolution layers (from 1/4 resolution down: fcn / fcn_skip's conv5 ... deconv3, the plain convolutions of unet /
 * res_unet) all slots in one launch (a page alone leaves them a partly filled chip); float32 engines run the pages one after
 * the other.  Each map equals pseg_predict_device's for that page.  Asynchronous
 * on `stream`.  The slots per unit (16, PSEG_BATCH_PAGES) are cut to what the device's free memory holds, and halved
 * again when an allocation fails all the same (PSEG_ENOMEM only when a single slot does not fit). */
int pseg_predict_pages_device(pseg_engine* e, const uint8_t* d_imgs, int n_pages, int H, int W, int64_t* d_labels,
                              uint8_t* d_labels_u8, void* stream);

/* Device-side error record of the engine -- the asynchronous `_device` entries cannot report what a kernel finds while it
 * runs.  Waits for `stream` (NULL: the engine's own), then reports AND clears: PSEG_OK, or PSEG_EHIP when one of the bounded
 * counter waits of the streamed-weights kernel (conv_sp_kernel: the 1/8-resolution layers of every page) gave up since the last
 * report -- the label maps produced since then are not valid (lib/network.py:256-259 has no such state: predict_on_batch either
 * returns the map or raises).  Every host-synchronous entry (pseg_predict, _batch, _chain, _exact_labels) runs the same check
 * before it returns; callers of pseg_predict_device / pseg_predict_pages_device call this where they synchronise. */
int pseg_engine_status(pseg_engine* e, void* stream);

/* Frees the activation tensors (all page slots) of the engine; the next predict call allocates what its page needs.  An
 * engine grows to the largest canvas x page-slot count it has seen (16 slots at 2048x1536: 14 GB) and keeps that; this is the
 * way back (tf.keras.backend.clear_session in lib/trainer.py:112 is the reference's).  Waits for the device. */
int pseg_engine_trim(pseg_engine* e);

/* Predictor.predict (lib/predictor.py:27-30): label maps of a list of pages of individual sizes.
 * The upload of page i+1 and the download of page i-1 overlap the compute of page i (two staging
 * slots, separate copy streams); runs of consecutive pages of one shape travel and compute as a unit (up to 8 pages,
 * pseg_predict_pages_device).  labels[i] (int64, H[i]*W[i]) and/or labels_u8[i]; either array may
 * be NULL, not both.  Host pointers; returns when every page is back. */
int pseg_predict_batch(pseg_engine* e, int n_pages, const uint8_t* const* imgs, const int* H,
                       const int* W, int64_t* const* labels, uint8_t* const* labels_u8);
/* How pseg_predict_batch cuts a page list into units (host logic, no device needed): runs of consecutive same-shape pages, at
 * most `cap` per unit, unit sizes 1, 2, 4 ... at the head of the list and ... 4, 2, 1 at its tail (the first upload and the last
 * download have no compute beside them).  Returns the number of units (>= 0) and fills unit_first / unit_count (each may be NULL),
 * or a negative PSEG_E*.  lib/predictor.py:27-30 has no such notion: the reference loops page by page. */
int pseg_batch_units(int n_pages, const int* H, const int* W, int cap, int* unit_first, int* unit_count, int max_units);

/* ---- Predictor chain: lib/predictor.py:32-54 ---------------------------------------------------------------- */

/* Predictor.predict_single / predict_masks as ONE device-resident call: Network.predict_single_data's argmax labels ->
 * [scale_to_original_shape (lib/output.py:63-79): order-0 resize of the label map to (Ho, Wo); Ho = 0: none] -> the
 * post-processors of PredictSettings.post_process in order (PSEG_POST_CC_VOTE = vote_connected_component_class,
 * PSEG_POST_BBOX = add_bounding_boxes; lib/postprocess.py:9-42) -> [generate_output_masks (lib/output.py:44-60)].
 * The uint8 label map stays in HBM between the stages; page and binarisation go up once, only the requested outputs
 * come down (DMA straight into page-locked caller memory).  img: uint8 (H,W[,in_channels]); binary: the ink map the vote
 * and the masks read, shape = the label map's final shape ((Ho,Wo) if resized, else (H,W)), may be NULL when neither is
 * asked for; flags: PSEG_CHAIN_EXACT_LABELS runs a bf16 engine's network stage in the label-exact mode.  Outputs (host,
 * each optional): labels int64 / labels_u8 (final shape), color / overlay / inverted / fg_color (final shape x 3).
 * Synchronous; <= 256 classes. */
enum { PSEG_POST_CC_VOTE = 1, PSEG_POST_BBOX = 2 };
enum { PSEG_CHAIN_EXACT_LABELS = 1 };
int pseg_predict_chain(pseg_engine* e, const uint8_t* img, int H, int W, int Ho, int Wo, const uint8_t* binary,
                       const int* post_ops, int n_post, unsigned flags, int64_t* labels, uint8_t* labels_u8,
                       const uint8_t* lut, int n_lut, uint8_t* color, uint8_t* overlay, uint8_t* inverted,
                       uint8_t* fg_color);

/* Label-exact throughput mode (lib/network.py:259: argmax of the float32 logits).  A bf16 engine's label map differs
 * from the float32 engine's only at near-ties of the two largest logits.  pseg_predict_margin_device runs the graph
 * and also writes the margin map (float32 (H,W): top-1 minus top-2 logit; labels_u8 optional).
 * pseg_predict_exact_labels_device aims at the float32 engine's label map at a fraction of its cost: bf16 pass +
 * margin, then the 32x32 blocks that hold a pixel with margin < tau are grouped into rectangles by a cost model and
 * re-evaluated, with their receptive-field halo, by a float32 companion engine (the bit-exact referee, same weights);
 * when the crops would cost more than one float32 pass over the page, the page goes through the float32 engine whole.
 * CALIBRATED, NOT PROVEN: tau starts at 4 x the bf16 path's measured logit error on three calibration crops and is
 * kept >= 2 x the largest change of a margin seen on ANY refereed pixel since the last weight change (every crop is
 * also a measurement; sentinel blocks are refereed on every page; an unflagged pixel that flips doubles tau) -- an
 * unflagged pixel outside every refereed rectangle is trusted on that evidence.  PSEG_MODE_F32_EXACT is the only mode
 * that is bit-exact by construction.  Synchronises `stream` internally (the flag map is read by the host).
 * d_labels (int64) and d_margin are optional outputs.  pseg_label_exact_stats: {tau, calibration logit error,
 * flagged pixel fraction, refereed block fraction, refereed area (with halos) / page area, tau escalations,
 * whole-page fallback (0/1), labels changed by the referee} of the last call; pseg_label_exact_stats_ex appends
 * {running margin error, rectangles refereed, cost-model price of the crops / price of the whole page, block edge,
 * 1 when the page went to the float32 engine directly -- after three whole-page referees in a row the next eight pages
 * skip the bf16 pass --} (`cap` doubles are written). */
int pseg_predict_margin_device(pseg_engine* e, const uint8_t* d_img, int H, int W, uint8_t* d_labels_u8,
                               float* d_margin, void* stream);
int pseg_predict_exact_labels_device(pseg_engine* e, const uint8_t* d_img, int H, int W, uint8_t* d_labels_u8,
                                     int64_t* d_labels, float* d_margin, void* stream);
/* Host-buffer form (Network.predict_single_data's `pred`, lib/network.py:259): either output may be NULL, not both. */
int pseg_predict_exact_labels(pseg_engine* e, const uint8_t* img, int H, int W, int64_t* labels, uint8_t* labels_u8);
int pseg_label_exact_stats(const pseg_engine* e, double out[8]);
int pseg_label_exact_stats_ex(const pseg_engine* e, double* out, int cap);

/* Page-locked host memory for the host-buffer entries ("uint8 page in pinned host memory -> label
 * map in host memory").  Pages and label maps that live in buffers from pseg_host_alloc, or in caller memory
 * registered with pseg_host_register, move by DMA straight between the caller's buffer and HBM, overlapped with
 * compute; any other (pageable) buffer is staged through the engine's two-slot pinned ring by the calling thread.
 * The reference keeps pages in NumPy arrays (lib/dataset.py:18-29); the Python shim offers pinned_empty(). */
int pseg_host_alloc(void** p, size_t bytes);
int pseg_host_free(void* p);
int pseg_host_register(void* p, size_t bytes);
int pseg_host_unregister(void* p);

/* Copy an intermediate activation to the host as float32 NHWC (true channel count) --
 * layer-by-layer parity tests.  `layer` is the Keras layer name.  dims[3] = {H,W,C} of the
 * padded canvas at that layer.  Valid after a predict call. */
int pseg_get_activation(pseg_engine* e, const char* layer, float* out, int64_t cap, int dims[3]);

/* The engine's own hipStream_t (what NULL stream arguments mean; the train step always runs on it): lets a caller
 * order its own work -- e.g. the data-parallel gradient all-reduce -- against the engine's kernels without
 * synchronising the device. */
void* pseg_engine_stream(pseg_engine* e);

/* Algorithmic forward FLOPs per canvas pixel (2 per MAC, true channel counts): the figure
 * SURVEY.md 8(d) quotes (113 700 for fcn_skip C=3). */
double pseg_flops_per_pixel(const pseg_engine* e);

/* Average duration in ms of the `slot`-th kernel class over the launches since the last
 * reset, measured with HIP events on the launch stream (bench.py roofline).  Timing is off
 * unless enabled; enabling inserts events around every launch. */
int pseg_timing_enable(pseg_engine* e, int on);
int pseg_timing_reset(pseg_engine* e);
int pseg_timing_num_slots(const pseg_engine* e);
int pseg_timing_get(pseg_engine* e, int slot, char* name, size_t name_cap, double* total_ms,
                    int64_t* launches, double* flops);

/* ---- Train: lib/network.py:90-104,167-246 (compile + fit), lib/metrics.py:8-17,60-85 ------- */

/* Training state of a PSEG_MODE_F32_EXACT engine (all four graphs, up to 64 classes): Keras-formulation Adam
 * (lib/architecture.py:83; beta1 .9, beta2 .999, eps 1e-7 are Keras' defaults) with per-tensor
 * clip-by-norm (`clipnorm`, lib/network.py:97; <= 0 disables) and optional clip-by-value. */
int pseg_train_init(pseg_engine* e, float beta1, float beta2, float eps, float clipnorm,
                    float clipvalue);

/* Optimizers (lib/architecture.py:71-90), each with Keras' TF 2.5 default hyper-parameters -- the
 * reference passes only lr / clipnorm / clipvalue (lib/network.py:92-102).  Default after
 * pseg_train_init: Adam.  Switching resets the optimizer state and the step count. */
enum { PSEG_OPT_ADAM = 0, PSEG_OPT_ADAMAX = 1, PSEG_OPT_ADADELTA = 2, PSEG_OPT_ADAGRAD = 3,
       PSEG_OPT_RMSPROP = 4, PSEG_OPT_SGD = 5, PSEG_OPT_NADAM = 6 };
int pseg_train_set_optimizer(pseg_engine* e, int optimizer);

/* Dropout (unet: lib/model.py:167,172, rate 0.5) is live in pseg_train_forward_backward* and the identity in
 * pseg_eval_step / pseg_predict*.  The mask of training forward number n (counted from this call) is a counter-based
 * function of (seed, n, layer, element): reproducible, different every step.  Default seed 0x1234. */
int pseg_train_set_dropout_seed(pseg_engine* e, uint32_t seed);

/* Loss functions (lib/metrics.py:8-9,72-133; `Loss` enum).  metrics[0] of the step calls is the selected
 * loss; the three other metrics stay accuracy / jacard_coef / dice_coef.  Default: cross-entropy. */
enum { PSEG_LOSS_CE = 0, PSEG_LOSS_JACCARD = 1, PSEG_LOSS_DICE = 2, PSEG_LOSS_HINGE = 3, PSEG_LOSS_FOCAL = 4,
       PSEG_LOSS_DICE_CE = 5 };
int pseg_train_set_loss(pseg_engine* e, int loss);

/* One sample (batch of one page, as the reference: lib/network.py:151-153): forward, mean sparse
 * softmax cross-entropy + metrics, backward.  img uint8 (H,W), mask uint8 class ids (H,W), host
 * pointers.  metrics = {loss, accuracy, jacard_coef, dice_coef} of this sample.  Gradients stay
 * in the engine's flat gradient buffer until pseg_train_apply. */
int pseg_train_forward_backward(pseg_engine* e, const uint8_t* img, const uint8_t* mask, int H,
                                int W, float metrics[4]);

/* Same with a float32 page on the 0..255 scale (an augmented sample, lib/network.py:149-161: the cubic warp
 * leaves non-integer values); the network input is img / 255.0f. */
int pseg_train_forward_backward_f32(pseg_engine* e, const float* img, const uint8_t* mask, int H, int W,
                                    float metrics[4]);

/* The flat device gradient buffer (all parameters in weight-table order, then the metric
 * accumulators): data-parallel training all-reduces exactly this buffer (one RCCL call), then
 * applies with grad_scale = 1/world. */
int pseg_train_grad_buffer(pseg_engine* e, float** d_grad, int64_t* count);
int pseg_train_metrics(pseg_engine* e, float metrics[4]);

/* Data-parallel training (the build's only collective; the reference trains in one process,
 * lib/network.py:235-241): one process per GPU, every rank runs pseg_train_forward_backward on its own page, then ONE
 * all-reduce(sum) of the flat gradient buffer over RCCL / xGMI, then pseg_train_apply(lr, 1/world) on every rank (clip
 * after averaging, identical updates).  RCCL is bound at run time (dlopen): libpseg.so does not link it.
 *   pseg_allreduce_unique_id: rank 0 creates the 128-byte communicator id; the caller hands it to the other ranks.
 *   pseg_allreduce_init:      ncclCommInitRank on the engine's device (collective: every rank calls it).
 *   pseg_train_allreduce:     the all-reduce, in place, enqueued on the engine's stream (no synchronisation).
 *   pseg_allreduce_destroy:   releases the communicator (pseg_destroy does it too). */
int pseg_allreduce_unique_id(uint8_t id[128]);
int pseg_allreduce_init(pseg_engine* e, int rank, int world, const uint8_t id[128]);
int pseg_train_allreduce(pseg_engine* e);
int pseg_allreduce_destroy(pseg_engine* e);
/* 1 when the build pinned the dlopen'ed RCCL entry points (ncclUniqueId size, ncclFloat32 / ncclSum values, the five prototypes)
 * against <rccl/rccl.h> with static_asserts, 0 when that header was absent at build time. */
int pseg_rccl_abi_pinned(void);

/* Clip + Adam update of every parameter with the (scaled) gradients; t += 1. */
int pseg_train_apply(pseg_engine* e, float lr, float grad_scale);

/* Gradient of one parameter in Keras layout (tests). */
int pseg_train_get_gradient(pseg_engine* e, const char* name, float* out, int64_t count);

/* model.evaluate step (lib/network.py:244-246): forward + loss/metrics only. */
int pseg_eval_step(pseg_engine* e, const uint8_t* img, const uint8_t* mask, int H, int W,
                   float metrics[4]);

/* ---- Post-process: lib/postprocess.py, lib/output.py ---------------------------------- */

/* vote_connected_component_class (lib/postprocess.py:9-26): 4-connected components of
 * `binary` (non-zero = ink); each takes its most frequent class in pred (ties -> lowest).
 * pred int64 (H,W) is updated in place, as the reference does. Host pointers. */
int pseg_cc_vote(int device, int64_t* pred, const uint8_t* binary, int H, int W, int n_classes);
/* Device variant: asynchronous on `stream`.  The label / histogram workspace is cached per device; calls from
 * different streams or threads are ordered against each other by an event (pseg_release_workspace frees it). */
int pseg_cc_vote_device(int device, int64_t* d_pred, const uint8_t* d_binary, int H, int W,
                        int n_classes, void* stream);

/* The same vote on the compact uint8 label map pseg_predict_device emits (n_classes <= 256): 4 instead of 25
 * algorithmic bytes per pixel; widen to int64 only at the host boundary. */
int pseg_cc_vote_device_u8(int device, uint8_t* d_pred, const uint8_t* d_binary, int H, int W,
                           int n_classes, void* stream);
/* Frees the device workspace the vote entries cache per device (waits for its last user). */
int pseg_release_workspace(int device);

/* add_bounding_boxes (lib/postprocess.py:29-42): every 4-connected component of each class
 * paints its bounding box; higher classes overwrite lower. out may alias nothing. */
int pseg_bbox_fill(int device, const int64_t* pred, int64_t* out, int H, int W, int n_classes);

/* add_bounding_boxes on the compact uint8 label map, device buffers (d_out must not alias d_pred); synchronises `stream`. */
int pseg_bbox_fill_device_u8(int device, const uint8_t* d_pred, uint8_t* d_out, int H, int W, int n_classes, void* stream);

/* generate_output_masks (lib/output.py:44-60).  lut: n_lut x 3 uint8 label->RGB
 * (ColorMap.to_rgb_array).  Outputs (H,W,3) uint8; any may be NULL. */
int pseg_masks(int device, const int64_t* pred, const uint8_t* binary, const uint8_t* lut,
               int n_lut, int H, int W, uint8_t* color, uint8_t* overlay, uint8_t* inverted,
               uint8_t* fg_color);
int pseg_masks_device(int device, const int64_t* d_pred, const uint8_t* d_binary,
                      const uint8_t* d_lut, int n_lut, int H, int W, uint8_t* d_color,
                      uint8_t* d_overlay, uint8_t* d_inverted, uint8_t* d_fg_color, void* stream);

/* generate_output_masks on the compact uint8 label map (14 instead of 21 algorithmic bytes per pixel); d_pred and
 * d_binary must be 4-byte aligned. */
int pseg_masks_device_u8(int device, const uint8_t* d_pred, const uint8_t* d_binary,
                         const uint8_t* d_lut, int n_lut, int H, int W, uint8_t* d_color,
                         uint8_t* d_overlay, uint8_t* d_inverted, uint8_t* d_fg_color, void* stream);

/* compute_char_height (lib/image_ops.py:58-82) minus the file read: Otsu threshold, invert
 * unless `inverse`, 4-connected components, keep glyph-shaped ones, upper median of heights.
 * *height = -1 when no component qualifies (the reference returns None). *otsu gets the
 * threshold (may be NULL). */
int pseg_otsu_char_height(int device, const uint8_t* gray, int H, int W, int inverse,
                          int* height, int* otsu);

/* ---- Line-height normalisation: lib/dataset.py:114-150, lib/util.py:21-29 ------------------ */

/* Output shape of skimage.transform.rescale as scale_binary calls it (lib/dataset.py:115):
 * np.round(scale * shape), half to even.  Host arithmetic only. */
int pseg_rescale_shape(int H, int W, double scale, int* Ho, int* Wo);

/* The anti-aliasing kernel scipy.ndimage.gaussian_filter builds for one axis: radius =
 * int(4 sigma + 0.5), w[i] = exp(-i^2 / 2 sigma^2) / sum, 2*radius+1 entries.  Uses libm's exp; a
 * caller that needs bit parity with a NumPy-based reference passes NumPy's kernel instead (the
 * two exp implementations can differ in the last bit).  w may be NULL to query the radius. */
int pseg_gaussian_kernel(double sigma, double* w, int cap, int* radius);

/* preserving_resize (lib/util.py:21-29) / scale_binary's gather (lib/dataset.py:114-119): order-0
 * warp of an (H,W) image of elem_bytes-sized pixels (1, 2, 3, 4 or 8) to (Ho,Wo).  Host pointers. */
int pseg_resize_nearest(int device, const void* src, int H, int W, int elem_bytes, void* dst,
                        int Ho, int Wo);

/* The same gather on device buffers, asynchronous on `stream` (scale_to_original_shape inside the Predictor chain). */
int pseg_resize_nearest_device(int device, const void* d_src, int H, int W, int elem_bytes, void* d_dst, int Ho, int Wo,
                               void* stream);

/* scale_image (lib/dataset.py:122-128): bicubic resize of a uint8 or float64 (H,W) plane to a
 * float64 (Ho,Wo) plane, clipped to the input range; Gaussian anti-aliasing (sigma = max(0,
 * (in/out - 1)/2) per axis) iff the image has more than two distinct values.  wy / wx: the
 * per-axis kernels with radii ry / rx (NULL: built with pseg_gaussian_kernel). */
int pseg_scale_image(int device, const void* src, int src_is_f64, int H, int W, double* dst,
                     int Ho, int Wo, const double* wy, int ry, const double* wx, int rx);

/* Augmentation warp (lib/data_generator.py / lib/network.py:149-161: keras-preprocessing's
 * apply_affine_transform -> scipy.ndimage.affine_transform(x, m, off, order, mode='nearest')): output pixel
 * (r, c) samples the input at m (r, c) + off; order 3 = cubic B-spline with prefilter (image), order 0 =
 * nearest (binary, mask).  float32 (H,W) planes, host pointers. */
int pseg_affine_warp(int device, const float* src, int H, int W, const double m[4], const double off[2],
                     int order, float* dst);
/* ... with the other fill mode the reference's AugmentationSettings name (lib/trainer.py:23-28 image_fill_mode / binary_fill_mode /
 * mask_fill_mode and *_cval -> keras-preprocessing apply_affine_transform(fill_mode, cval)): fill_mode 0 'nearest', 1 'constant'
 * (scipy mode='constant': cval where the source coordinate leaves [0, n - 1]; the spline prefilter runs on the unpadded plane),
 * 2 'reflect' (d c b a | a b c d | d c b a; half-sample-symmetric prefilter), 3 'wrap' (scipy's legacy 'wrap': period n - 1,
 * mirror prefilter) -- the four values keras-preprocessing accepts; semantics of scipy >= 1.6's map_coordinate. */
int pseg_affine_warp_fill(int device, const float* src, int H, int W, const double m[4], const double off[2],
                          int order, int fill_mode, float cval, float* dst);
/* AugmentationSettings.brightness_range (lib/trainer.py:21,33; removed from the binary / mask generators at :45,50): the image
 * generator draws one factor per sample and keras-preprocessing 1.1.2 applies apply_brightness_shift(x, factor, scale=False) --
 * through an 8-bit PIL image: planes outside [0, 255] are stretched to it first and mapped back afterwards, the gain is
 * PIL's ImageEnhance.Brightness (truncating blend with black, clipped).  src / dst: float32 planes of n values (all channels of
 * the sample: min / max are taken over the whole array), host pointers; dst may equal src. */
int pseg_brightness_shift(int device, const float* src, int64_t n, float brightness, float* dst);

/* ---- evaluation reductions ------------------------------------------------------------ */

/* Joint histogram behind fgpa / fgoverlap_per_class (lib/image_ops.py:8-55) and count_matches /
 * total_accuracy (lib/evaluation.py:8-32): counts[b][m][p] = pixels with (binary != 0) == b, mask label m,
 * predicted label p; (n_classes + 1) slots per axis, the last one collects labels outside [0, n_classes).
 * pred / mask: n labels of pred_bytes / mask_bytes (1, 4 or 8; signed for 4 and 8) each; binary may be NULL
 * (every pixel counts as ink).  counts: 2 * (n_classes+1)^2 int64.  Host pointers. */
int pseg_eval_confusion(int device, const void* pred, int pred_bytes, const void* mask, int mask_bytes,
                        const uint8_t* binary, int64_t n, int n_classes, int64_t* counts);

/* cv2.connectedComponentsWithStats(binary, connectivity)'s label image (lib/evaluation.py:84-85): 0 = paper,
 * components numbered 1.. in the order OpenCV's scan meets them (connectivity 4: raster order of the first
 * pixel; 8: raster order of the first 2x2 block).  labels: int32 (H,W); *num_labels counts the background. */
int pseg_cc_label(int device, const uint8_t* binary, int H, int W, int connectivity, int32_t* labels,
                  int32_t* num_labels);

/* Per-component tables for ConnectedComponentEval (lib/evaluation.py:73-117); every output may be NULL.
 * stats: int32 [num_labels][5] = cv2's LEFT, TOP, WIDTH, HEIGHT, AREA; centroids: double [num_labels][2]
 * (x, y); eq[l] = pixels of l with pred == mask; hist_pred / hist_mask: [num_labels][n_classes+1] class
 * counts (last slot: out of range); order: the H*W pixel indices sorted by (label, raster position), so
 * that `bbox(image)[component]` (lib/evaluation.py:104-106) is image.ravel()[order[o:o+area]]. */
int pseg_cc_tables(int device, const int32_t* labels, int H, int W, int num_labels, const void* pred,
                   int pred_bytes, const void* mask, int mask_bytes, int n_classes, int32_t* stats,
                   double* centroids, int64_t* eq, int64_t* hist_pred, int64_t* hist_mask, int32_t* order);

/* prepare_images (lib/dataset.py:131-150), all pixel work on the device in one call.
 * image, binary: uint8 (H0,W0) scan and binarisation (paper = 1 or 255).  (H1,W1) =
 * pseg_rescale_shape(H0, W0, target_line_height / line_height_px); (H2,W2) = the max_width stage
 * (pseg_rescale_shape(H1, W1, max_width / W1) when that factor is < 1), or 0,0 for none.
 * Outputs: out_img uint8 network input (ink bright), out_bin uint8 ink = 1, both of the final
 * shape; out_orig_bin (H0,W0) ink map (may be NULL); out_stage1 float64 (H1,W1) bicubic result
 * before inversion (may be NULL; tests). */
int pseg_prepare_images(int device, const uint8_t* image, const uint8_t* binary, int H0, int W0,
                        int H1, int W1, const double* wy1, int ry1, const double* wx1, int rx1,
                        int H2, int W2, const double* wy2, int ry2, const double* wx2, int rx2,
                        uint8_t* out_img, uint8_t* out_bin, uint8_t* out_orig_bin,
                        double* out_stage1);

#ifdef __cplusplus
}
#endif
#endif /* PSEG_H */
