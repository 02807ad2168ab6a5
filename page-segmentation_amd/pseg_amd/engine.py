"""ctypes binding of include/pseg.h.  No compute happens in Python; there is no CPU fallback:
every call raises PsegError when libpseg.so or a HIP device is missing."""
import ctypes
import os

import numpy as np

_CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csrc")

ARCH_IDS = {"fcn_skip": 0, "fcn": 1, "unet": 2, "res_unet": 3}
MODE_F32_EXACT = 0
MODE_BF16 = 1
FLAG_BATCHNORM = 1

# every symbol include/pseg.h declares (tests check that the library exports each one)
EXPORTED_SYMBOLS = (
    "pseg_abi_version", "pseg_last_error", "pseg_device_count", "pseg_create", "pseg_create_ex", "pseg_create_plan", "pseg_env_knobs", "pseg_destroy",
    "pseg_num_weights", "pseg_weight_info", "pseg_set_weights", "pseg_get_weights",
    "pseg_predict", "pseg_predict_device", "pseg_predict_pages_device", "pseg_engine_status", "pseg_engine_trim", "pseg_batch_units", "pseg_rccl_abi_pinned", "pseg_predict_batch", "pseg_predict_chain", "pseg_get_activation", "pseg_flops_per_pixel", "pseg_engine_stream",
    "pseg_host_alloc", "pseg_host_free", "pseg_host_register", "pseg_host_unregister",
    "pseg_predict_margin_device", "pseg_predict_exact_labels_device", "pseg_predict_exact_labels", "pseg_label_exact_stats", "pseg_label_exact_stats_ex",
    "pseg_timing_enable", "pseg_timing_reset", "pseg_timing_num_slots", "pseg_timing_get",
    "pseg_train_init", "pseg_train_set_optimizer", "pseg_train_set_loss", "pseg_train_set_dropout_seed", "pseg_train_forward_backward", "pseg_train_forward_backward_f32", "pseg_train_grad_buffer", "pseg_train_metrics",
    "pseg_train_apply", "pseg_train_get_gradient", "pseg_eval_step",
    "pseg_allreduce_unique_id", "pseg_allreduce_init", "pseg_train_allreduce", "pseg_allreduce_destroy",
    "pseg_cc_vote", "pseg_cc_vote_device", "pseg_cc_vote_device_u8", "pseg_release_workspace", "pseg_bbox_fill",
    "pseg_masks", "pseg_masks_device", "pseg_masks_device_u8", "pseg_bbox_fill_device_u8",
    "pseg_otsu_char_height",
    "pseg_rescale_shape", "pseg_gaussian_kernel", "pseg_resize_nearest", "pseg_resize_nearest_device", "pseg_scale_image",
    "pseg_prepare_images", "pseg_affine_warp", "pseg_affine_warp_fill", "pseg_brightness_shift",
    "pseg_eval_confusion", "pseg_cc_label", "pseg_cc_tables",
)


PLAN_FROM_ENV = bool(os.environ.get("PSEG_PLAN_FROM_ENV"))   # test harness / tools: PSEG_* of os.environ -> plan switches of new engines (see Engine)


class PsegError(Exception):
    """The reference raises bare Exception(...) (lib/dataset.py:67, lib/trainer.py:133)."""


_LIB = None


def lib_path():
    # PSEG_LIB selects another build of the same ABI (e.g. csrc/libpseg_diag.so with trace stamps)
    return os.environ.get("PSEG_LIB") or os.path.join(_CSRC, "libpseg.so")


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise PsegError("libpseg.so is not built (%s); run __graft_entry__.build() -- there is "
                        "no CPU fallback" % path)
    L = ctypes.CDLL(path)
    c = ctypes
    vp, i, i64 = c.c_void_p, c.c_int, c.c_int64
    L.pseg_last_error.restype = c.c_char_p
    L.pseg_create.argtypes = [i, i, i, i, i, c.POINTER(vp)]
    L.pseg_create_ex.argtypes = [i, i, i, i, i, c.c_uint, c.POINTER(vp)]
    L.pseg_create_plan.argtypes = [i, i, i, i, i, c.c_uint, c.c_char_p, c.POINTER(vp)]
    L.pseg_env_knobs.restype = c.c_char_p
    L.pseg_destroy.argtypes = [vp]
    L.pseg_num_weights.argtypes = [vp]
    L.pseg_weight_info.argtypes = [vp, i, c.c_char_p, c.c_size_t, c.POINTER(i64), c.POINTER(i)]
    L.pseg_set_weights.argtypes = [vp, c.c_char_p, vp, c.POINTER(i64), i]
    L.pseg_get_weights.argtypes = [vp, c.c_char_p, vp, i64]
    L.pseg_predict.argtypes = [vp, vp, i, i, vp, vp, vp]
    L.pseg_predict_device.argtypes = [vp, vp, i, i, vp, vp, vp, vp, vp]
    L.pseg_predict_pages_device.argtypes = [vp, vp, i, i, i, vp, vp, vp]
    L.pseg_engine_status.argtypes = [vp, vp]
    L.pseg_engine_trim.argtypes = [vp]
    L.pseg_predict_batch.argtypes = [vp, i, vp, vp, vp, vp, vp]
    L.pseg_predict_chain.argtypes = [vp, vp, i, i, i, i, vp, c.POINTER(i), i, c.c_uint, vp, vp, vp, i, vp, vp, vp, vp]
    L.pseg_bbox_fill_device_u8.argtypes = [i, vp, vp, i, i, i, vp]
    L.pseg_resize_nearest_device.argtypes = [i, vp, i, i, i, vp, i, i, vp]
    L.pseg_predict_margin_device.argtypes = [vp, vp, i, i, vp, vp, vp]
    L.pseg_predict_exact_labels_device.argtypes = [vp, vp, i, i, vp, vp, vp, vp]
    L.pseg_predict_exact_labels.argtypes = [vp, vp, i, i, vp, vp]
    L.pseg_label_exact_stats.argtypes = [vp, c.POINTER(c.c_double)]
    L.pseg_label_exact_stats_ex.argtypes = [vp, c.POINTER(c.c_double), i]
    L.pseg_host_alloc.argtypes = [c.POINTER(vp), c.c_size_t]
    L.pseg_host_free.argtypes = [vp]
    L.pseg_host_register.argtypes = [vp, c.c_size_t]
    L.pseg_host_unregister.argtypes = [vp]
    L.pseg_get_activation.argtypes = [vp, c.c_char_p, vp, i64, c.POINTER(i)]
    L.pseg_engine_stream.argtypes = [vp]
    L.pseg_engine_stream.restype = vp
    L.pseg_flops_per_pixel.argtypes = [vp]
    L.pseg_flops_per_pixel.restype = c.c_double
    L.pseg_timing_enable.argtypes = [vp, i]
    L.pseg_timing_reset.argtypes = [vp]
    L.pseg_timing_num_slots.argtypes = [vp]
    L.pseg_timing_get.argtypes = [vp, i, c.c_char_p, c.c_size_t, c.POINTER(c.c_double),
                                  c.POINTER(i64), c.POINTER(c.c_double)]
    f = c.c_float
    L.pseg_train_init.argtypes = [vp, f, f, f, f, f]
    L.pseg_train_set_optimizer.argtypes = [vp, i]
    L.pseg_train_set_loss.argtypes = [vp, i]
    L.pseg_train_set_dropout_seed.argtypes = [vp, c.c_uint32]
    L.pseg_train_forward_backward.argtypes = [vp, vp, vp, i, i, c.POINTER(f)]
    L.pseg_train_forward_backward_f32.argtypes = [vp, vp, vp, i, i, c.POINTER(f)]
    L.pseg_train_grad_buffer.argtypes = [vp, c.POINTER(vp), c.POINTER(i64)]
    L.pseg_train_metrics.argtypes = [vp, c.POINTER(f)]
    L.pseg_train_apply.argtypes = [vp, f, f]
    L.pseg_train_get_gradient.argtypes = [vp, c.c_char_p, vp, i64]
    L.pseg_eval_step.argtypes = [vp, vp, vp, i, i, c.POINTER(f)]
    L.pseg_allreduce_unique_id.argtypes = [vp]
    L.pseg_allreduce_init.argtypes = [vp, i, i, vp]
    L.pseg_train_allreduce.argtypes = [vp]
    L.pseg_allreduce_destroy.argtypes = [vp]
    L.pseg_cc_vote.argtypes = [i, vp, vp, i, i, i]
    d = c.c_double
    L.pseg_rescale_shape.argtypes = [i, i, d, c.POINTER(i), c.POINTER(i)]
    L.pseg_gaussian_kernel.argtypes = [d, vp, i, c.POINTER(i)]
    L.pseg_resize_nearest.argtypes = [i, vp, i, i, i, vp, i, i]
    L.pseg_scale_image.argtypes = [i, vp, i, i, i, vp, i, i, vp, i, vp, i]
    L.pseg_prepare_images.argtypes = [i, vp, vp, i, i, i, i, vp, i, vp, i, i, i, vp, i, vp, i, vp, vp, vp, vp]
    L.pseg_affine_warp.argtypes = [i, vp, i, i, vp, vp, i, vp]
    L.pseg_affine_warp_fill.argtypes = [i, vp, i, i, vp, vp, i, i, ctypes.c_float, vp]
    L.pseg_brightness_shift.argtypes = [i, vp, i64, ctypes.c_float, vp]
    L.pseg_cc_vote_device.argtypes = [i, vp, vp, i, i, i, vp]
    L.pseg_cc_vote_device_u8.argtypes = [i, vp, vp, i, i, i, vp]
    L.pseg_release_workspace.argtypes = [i]
    L.pseg_masks_device_u8.argtypes = [i, vp, vp, vp, i, i, i, vp, vp, vp, vp, vp]
    L.pseg_eval_confusion.argtypes = [i, vp, i, vp, i, vp, i64, i, vp]
    L.pseg_cc_label.argtypes = [i, vp, i, i, i, vp, c.POINTER(c.c_int32)]
    L.pseg_cc_tables.argtypes = [i, vp, i, i, i, vp, i, vp, i, i, vp, vp, vp, vp, vp, vp]
    L.pseg_bbox_fill.argtypes = [i, vp, vp, i, i, i]
    L.pseg_masks.argtypes = [i, vp, vp, vp, i, i, i, vp, vp, vp, vp]
    L.pseg_masks_device.argtypes = [i, vp, vp, vp, i, i, i, vp, vp, vp, vp, vp]
    L.pseg_otsu_char_height.argtypes = [i, vp, i, i, i, c.POINTER(i), c.POINTER(i)]
    _LIB = L
    return L


def _check(rc):
    if rc != 0:
        raise PsegError(lib().pseg_last_error().decode("utf-8", "replace") or "pseg error %d" % rc)


def device_count():
    return int(lib().pseg_device_count())


def _ptr(a):
    return None if a is None else ctypes.c_void_p(a.ctypes.data)


class _PinnedBlock:
    """Owner of one pseg_host_alloc block; freed when the last array viewing it goes away."""

    def __init__(self, nbytes):
        p = ctypes.c_void_p()
        _check(lib().pseg_host_alloc(ctypes.byref(p), int(nbytes)))
        self.ptr, self.nbytes = p.value, int(nbytes)

    def __del__(self):
        try:
            if self.ptr:
                lib().pseg_host_free(ctypes.c_void_p(self.ptr))
                self.ptr = None
        except Exception:
            pass


def pinned_empty(shape, dtype=np.uint8):
    """NumPy array in page-locked host memory (pseg_host_alloc): pages and label maps kept in such arrays move by DMA
    straight from / to the array in Engine.predict_batch, overlapped with compute."""
    dt = np.dtype(dtype)
    n = int(np.prod(shape, dtype=np.int64)) * dt.itemsize
    blk = _PinnedBlock(max(n, 1))
    buf = (ctypes.c_char * max(n, 1)).from_address(blk.ptr)
    buf._pseg_block = blk                       # keeps the block alive as long as any view of the buffer
    a = np.frombuffer(buf, dtype=dt, count=int(np.prod(shape, dtype=np.int64))).reshape(shape)
    return a


class _PinnedPool:
    """Recycled page-locked blocks for arrays handed to callers (the Predictor chain's label maps and masks): pinning
    150 MB per call costs more than the chain itself, and a fresh pageable array that size pays ~18 ms of first-touch
    page faults before the copy starts.  A block returns to the pool when the last NumPy view of it is collected."""
    MAX_BYTES = 2 << 30

    def __init__(self):
        self.free, self.held = {}, 0

    def take(self, nbytes):
        lst = self.free.get(nbytes)
        if lst:
            self.held -= nbytes
            return lst.pop()
        return _PinnedBlock(nbytes)

    def give(self, blk):
        if self.held + blk.nbytes > self.MAX_BYTES:
            return                                   # dropped: _PinnedBlock.__del__ frees it
        self.free.setdefault(blk.nbytes, []).append(blk)
        self.held += blk.nbytes

    def clear(self):
        self.free, self.held = {}, 0


_POOL = _PinnedPool()


class _PooledLease:
    def __init__(self, blk):
        self.blk = blk

    def __del__(self):
        try:
            _POOL.give(self.blk)
        except Exception:
            pass


def pinned_empty_pooled(shape, dtype=np.uint8):
    """As pinned_empty, from the recycling pool (sizes rounded up to 1 MiB so that pages of one format share blocks)."""
    dt = np.dtype(dtype)
    count = int(np.prod(shape, dtype=np.int64))
    n = max(count * dt.itemsize, 1)
    n = (n + (1 << 20) - 1) >> 20 << 20
    blk = _POOL.take(n)
    buf = (ctypes.c_char * n).from_address(blk.ptr)
    buf._pseg_lease = _PooledLease(blk)
    return np.frombuffer(buf, dtype=dt, count=count).reshape(shape)


def pinned_copy(a):
    """A page-locked copy of `a`."""
    out = pinned_empty(a.shape, a.dtype)
    out[...] = a
    return out


class Engine:
    """Opaque pseg_engine handle: one FCN graph + its weights resident on one GPU."""

    def __init__(self, arch="fcn_skip", n_classes=3, in_channels=1, device=0, mode=MODE_BF16, batch_norm=False, plan=None):
        """batch_norm: BatchNormalization at res_unet's bn_act sites (lib/model.py:265-271; PSEG_FLAG_BATCHNORM).
        plan: plan switches for pseg_create_plan, "PSEG_NO_DQ=1;PSEG_WS_FORM=2" or a dict (tests and A/B measurements: every
        switch keeps the results within the documented bars).  With PLAN_FROM_ENV set (tests/conftest.py, tools/) and no plan given,
        the PSEG_* variables of os.environ that the library does not read itself become the plan -- the test suite keeps steering
        kernels with monkeypatch.setenv while the release library ignores the environment beyond pseg_env_knobs()."""
        self._h = None
        L = lib()
        arch_id = ARCH_IDS[arch] if isinstance(arch, str) else int(arch)
        h = ctypes.c_void_p()
        if plan is None and PLAN_FROM_ENV:
            listed = set(L.pseg_env_knobs().decode().split("\n"))
            plan = {k: v for k, v in os.environ.items() if k.startswith("PSEG_") and k not in listed and k not in ("PSEG_LIB",)}
        if isinstance(plan, dict):
            plan = ";".join("%s=%s" % kv for kv in sorted(plan.items()))
        _check(L.pseg_create_plan(arch_id, int(n_classes), int(in_channels), int(device), int(mode),
                                  FLAG_BATCHNORM if batch_norm else 0, (plan or "").encode(), ctypes.byref(h)))
        self._h = h
        self.batch_norm = bool(batch_norm)
        self.arch = arch
        self.n_classes = int(n_classes)
        self.in_channels = int(in_channels)
        self.device = int(device)
        self.mode = int(mode)

    def close(self):
        if self._h is not None:
            lib().pseg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- weights -------------------------------------------------------------------------------
    def weight_specs(self):
        L = lib()
        out = []
        name = ctypes.create_string_buffer(128)
        shape = (ctypes.c_int64 * 4)()
        nd = ctypes.c_int()
        for i in range(L.pseg_num_weights(self._h)):
            _check(L.pseg_weight_info(self._h, i, name, 128, shape, ctypes.byref(nd)))
            out.append((name.value.decode(), tuple(int(shape[k]) for k in range(nd.value))))
        return out

    def set_weights(self, weights):
        """weights: mapping name -> ndarray in Keras layout."""
        L = lib()
        for name, arr in weights.items():
            a = np.ascontiguousarray(arr, dtype=np.float32)
            shp = (ctypes.c_int64 * max(a.ndim, 1))(*a.shape)
            _check(L.pseg_set_weights(self._h, name.encode(), _ptr(a), shp, a.ndim))

    def get_weights(self):
        L = lib()
        out = {}
        for name, shp in self.weight_specs():
            a = np.empty(shp, np.float32)
            _check(L.pseg_get_weights(self._h, name.encode(), _ptr(a), a.size))
            out[name] = a
        return out

    # -- predict -------------------------------------------------------------------------------
    def predict(self, image, want_logits=True, want_probs=True, want_labels=True):
        """uint8 (H,W) [or (H,W,3)] -> (logits f32 (H,W,C) | None, probs | None, labels int64 | None).
        Mirrors Network.predict_single_data (lib/network.py:248-260)."""
        img = np.ascontiguousarray(image, dtype=np.uint8)
        if img.ndim == 3 and img.shape[2] == 1:
            img = img[..., 0]
        if img.ndim != (2 if self.in_channels == 1 else 3):
            raise PsegError("image of shape %r does not match in_channels=%d" % (img.shape, self.in_channels))
        H, W = img.shape[:2]
        C = self.n_classes
        logits = np.empty((H, W, C), np.float32) if want_logits else None
        probs = np.empty((H, W, C), np.float32) if want_probs else None
        labels = np.empty((H, W), np.int64) if want_labels else None
        _check(lib().pseg_predict(self._h, _ptr(img), H, W, _ptr(logits), _ptr(probs), _ptr(labels)))
        return logits, probs, labels

    def predict_device(self, d_img, H, W, d_logits=0, d_probs=0, d_labels=0, d_labels_u8=0, stream=0):
        """Raw device pointers (ints, e.g. torch.Tensor.data_ptr()); asynchronous."""
        _check(lib().pseg_predict_device(self._h, ctypes.c_void_p(d_img), int(H), int(W),
                                         ctypes.c_void_p(d_logits or None), ctypes.c_void_p(d_probs or None),
                                         ctypes.c_void_p(d_labels or None), ctypes.c_void_p(d_labels_u8 or None),
                                         ctypes.c_void_p(stream or None)))

    def predict_pages_device(self, d_imgs, n_pages, H, W, d_labels=0, d_labels_u8=0, stream=0):
        """n_pages pages of one shape, contiguous on the device -> their label maps, contiguous (lib/predictor.py:27-30); asynchronous."""
        _check(lib().pseg_predict_pages_device(self._h, ctypes.c_void_p(d_imgs), int(n_pages), int(H), int(W), ctypes.c_void_p(d_labels or None),
                                               ctypes.c_void_p(d_labels_u8 or None), ctypes.c_void_p(stream or None)))

    def status(self, stream=0):
        """Waits for `stream` (0: the engine's own), then raises PsegError if a kernel reported an error since the last check
        (pseg_engine_status): what the asynchronous *_device calls cannot tell their caller."""
        _check(lib().pseg_engine_status(self._h, ctypes.c_void_p(stream or None)))

    def trim(self):
        """Frees the activation tensors (every page slot); the next predict call allocates what it needs (pseg_engine_trim)."""
        _check(lib().pseg_engine_trim(self._h))

    def predict_margin_device(self, d_img, H, W, d_margin, d_labels_u8=0, stream=0):
        """As predict_device, plus the float32 (H,W) margin map: top-1 minus top-2 logit per pixel."""
        _check(lib().pseg_predict_margin_device(self._h, ctypes.c_void_p(d_img), int(H), int(W), ctypes.c_void_p(d_labels_u8 or None),
                                                ctypes.c_void_p(d_margin), ctypes.c_void_p(stream or None)))

    def predict_exact_labels_device(self, d_img, H, W, d_labels_u8, d_labels=0, d_margin=0, stream=0):
        """Label-exact throughput mode: bf16 pass + margin map, float32 referee on the blocks that hold near-ties; the
        uint8 label map is the float32 engine's wherever the referee looked and wherever the bf16 margin exceeds the
        calibrated threshold (calibrated, not proven: mode=MODE_F32_EXACT is the only bit-exact mode).  Synchronises the stream."""
        _check(lib().pseg_predict_exact_labels_device(self._h, ctypes.c_void_p(d_img), int(H), int(W), ctypes.c_void_p(d_labels_u8),
                                                      ctypes.c_void_p(d_labels or None), ctypes.c_void_p(d_margin or None),
                                                      ctypes.c_void_p(stream or None)))

    def predict_exact_labels(self, image, dtype=np.int64):
        """uint8 (H,W) page -> label map equal to the float32 engine's (host arrays; label-exact mode)."""
        img = np.ascontiguousarray(image, dtype=np.uint8)
        H, W = img.shape[:2]
        out = np.empty((H, W), dtype)
        if np.dtype(dtype) == np.int64:
            _check(lib().pseg_predict_exact_labels(self._h, _ptr(img), H, W, _ptr(out), None))
        elif np.dtype(dtype) == np.uint8:
            _check(lib().pseg_predict_exact_labels(self._h, _ptr(img), H, W, None, _ptr(out)))
        else:
            raise PsegError("labels dtype must be int64 or uint8")
        return out

    def label_exact_stats(self):
        """Statistics of the last predict_exact_labels[_device] call."""
        v = (ctypes.c_double * 13)()
        _check(lib().pseg_label_exact_stats_ex(self._h, v, 13))
        return {"tau": float(v[0]), "calib_logit_err": float(v[1]), "flagged_px_frac": float(v[2]), "referee_tile_frac": float(v[3]),
                "referee_area_frac": float(v[4]), "tau_escalations": int(v[5]), "whole_page_fallback": int(v[6]),
                "labels_changed": int(v[7]), "margin_err_running": float(v[8]), "referee_rects": int(v[9]),
                "referee_cost_vs_full_page": float(v[10]), "flag_block": int(v[11]), "direct_float32": int(v[12])}

    POST_OPS = {"cc_vote": 1, "bbox": 2}

    def predict_chain(self, image, binary=None, out_shape=None, post_ops=(), exact_labels=False, labels="u8", lut=None,
                      masks=False):
        """The Predictor chain on the device (pseg_predict_chain; lib/predictor.py:32-54): predict -> [nearest resize of
        the label map to out_shape] -> post-processors ("cc_vote" / "bbox", in order) -> [the four masks].  `binary` must
        have the label map's final shape.  Returns {"labels": uint8 or int64 map or None, "masks": (color, overlay,
        inverted, fg_color) or None}; the arrays live in recycled page-locked memory."""
        img = np.ascontiguousarray(image, dtype=np.uint8)
        H, W = img.shape[:2]
        Ho, Wo = (int(out_shape[0]), int(out_shape[1])) if out_shape is not None else (0, 0)
        Hl, Wl = (Ho, Wo) if Ho > 0 else (H, W)
        b = None
        if binary is not None:
            b = np.ascontiguousarray(binary, dtype=np.uint8)
            if b.shape != (Hl, Wl):
                raise PsegError("binary has shape %r, the label map %r" % (b.shape, (Hl, Wl)))
        ops = (ctypes.c_int * max(len(post_ops), 1))(*[self.POST_OPS[o] if isinstance(o, str) else int(o) for o in post_ops])
        lab = None
        if labels == "u8":
            lab = pinned_empty_pooled((Hl, Wl), np.uint8)
        elif labels == "i64":
            lab = pinned_empty_pooled((Hl, Wl), np.int64)
        elif labels is not None:
            raise PsegError("labels must be 'u8', 'i64' or None")
        t = None
        outs = [None] * 4
        if masks:
            t = np.ascontiguousarray(lut, dtype=np.uint8).reshape(-1, 3)
            outs = [pinned_empty_pooled((Hl, Wl, 3), np.uint8) for _ in range(4)]
        _check(lib().pseg_predict_chain(self._h, _ptr(img), H, W, Ho, Wo, _ptr(b), ops, len(post_ops), 1 if exact_labels else 0,
                                        _ptr(lab) if labels == "i64" else None, _ptr(lab) if labels == "u8" else None,
                                        _ptr(t), 0 if t is None else t.shape[0], *[_ptr(o) for o in outs]))
        return {"labels": lab, "masks": tuple(outs) if masks else None}

    def predict_batch(self, images, dtype=np.int64, out=None):
        """Label maps of a list of (H,W) uint8 pages (sizes may differ); copies overlap compute.
        `out`: optional list of preallocated C-contiguous label arrays to fill (fresh 25 MB arrays cost
        more in first-touch page faults than the transfer itself)."""
        imgs = [np.ascontiguousarray(im, dtype=np.uint8) for im in images]
        n = len(imgs)
        if any(im.ndim != 2 for im in imgs) and self.in_channels == 1:
            raise PsegError("pages must be 2-D uint8 arrays")
        if out is not None:
            if len(out) != n or any(o.shape != im.shape[:2] or o.dtype != np.dtype(dtype) or not o.flags.c_contiguous
                                    for o, im in zip(out, imgs)):
                raise PsegError("out must hold one C-contiguous %s array of the page's shape per page" % np.dtype(dtype))
            outs = list(out)
        else:
            outs = [np.empty(im.shape[:2], dtype) for im in imgs]
        P = ctypes.c_void_p * max(n, 1)
        I = ctypes.c_int * max(n, 1)
        ip = P(*[im.ctypes.data for im in imgs]) if n else P()
        op = P(*[o.ctypes.data for o in outs]) if n else P()
        hs, ws = I(*[im.shape[0] for im in imgs]), I(*[im.shape[1] for im in imgs])
        if np.dtype(dtype) == np.int64:
            _check(lib().pseg_predict_batch(self._h, n, ip, hs, ws, op, None))
        elif np.dtype(dtype) == np.uint8:
            _check(lib().pseg_predict_batch(self._h, n, ip, hs, ws, None, op))
        else:
            raise PsegError("labels dtype must be int64 or uint8")
        return outs

    def activation(self, layer):
        dims = (ctypes.c_int * 3)()
        _check(lib().pseg_get_activation(self._h, layer.encode(), None, 0, dims))
        out = np.empty((dims[0], dims[1], dims[2]), np.float32)
        _check(lib().pseg_get_activation(self._h, layer.encode(), _ptr(out), out.size, dims))
        return out

    # -- training (float32 engine) -----------------------------------------------------------------
    def train_init(self, beta1=0.9, beta2=0.999, eps=1e-7, clipnorm=1.0, clipvalue=0.0):
        """Keras Adam defaults; clipnorm is per tensor (lib/network.py:97), <= 0 disables."""
        _check(lib().pseg_train_init(self._h, beta1, beta2, eps, clipnorm, clipvalue))

    # -- data-parallel training through the C ABI (RCCL bound at run time; pseg_amd.parallel offers torch.distributed instead)
    @staticmethod
    def allreduce_unique_id():
        buf = (ctypes.c_uint8 * 128)()
        _check(lib().pseg_allreduce_unique_id(buf))
        return bytes(buf)

    def allreduce_init(self, rank, world, unique_id):
        buf = (ctypes.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        _check(lib().pseg_allreduce_init(self._h, int(rank), int(world), buf))

    def train_allreduce(self):
        _check(lib().pseg_train_allreduce(self._h))

    def allreduce_destroy(self):
        _check(lib().pseg_allreduce_destroy(self._h))

    OPTIMIZERS = {"adam": 0, "adamax": 1, "adadelta": 2, "adagrad": 3, "rmsprop": 4, "sgd": 5, "nadam": 6}

    def train_set_optimizer(self, name):
        """One of the reference's Optimizers enum values (lib/architecture.py:71-78), Keras defaults."""
        if name not in self.OPTIMIZERS:
            raise PsegError("unknown optimizer %r" % (name,))
        _check(lib().pseg_train_set_optimizer(self._h, self.OPTIMIZERS[name]))

    LOSSES = {"categorical_crossentropy": 0, "jaccard": 1, "dice": 2, "categorical_hinge": 3, "categorical_focal": 4,
              "dice_and_crossentropy": 5}

    def train_set_loss(self, name):
        """One of the reference's Loss enum values (lib/metrics.py:116-121)."""
        if name not in self.LOSSES:
            raise PsegError("unknown loss %r" % (name,))
        _check(lib().pseg_train_set_loss(self._h, self.LOSSES[name]))

    def _img_mask(self, image, mask):
        img = np.ascontiguousarray(image, dtype=np.uint8)
        msk = np.ascontiguousarray(mask, dtype=np.uint8)
        if img.shape[:2] != msk.shape[:2]:
            raise PsegError("image %r and mask %r differ in shape" % (img.shape, msk.shape))
        return img, msk

    def train_forward_backward(self, image, mask):
        """-> (loss, accuracy, jacard_coef, dice_coef) of this page; gradients stay on the device."""
        img, msk = self._img_mask(image, mask)
        m = (ctypes.c_float * 4)()
        _check(lib().pseg_train_forward_backward(self._h, _ptr(img), _ptr(msk), img.shape[0], img.shape[1], m))
        return tuple(float(v) for v in m)

    def train_set_dropout_seed(self, seed):
        """Seed of the Dropout masks (unet); also restarts the step counter the masks depend on."""
        _check(lib().pseg_train_set_dropout_seed(self._h, int(seed) & 0xFFFFFFFF))

    def train_forward_backward_float(self, image, mask):
        """As train_forward_backward with a float32 page on the 0..255 scale (augmented sample)."""
        img = np.ascontiguousarray(image, dtype=np.float32)
        m = np.ascontiguousarray(mask, dtype=np.uint8)
        if img.shape[:2] != m.shape[:2]:
            raise PsegError("image %r and mask %r differ in shape" % (img.shape, m.shape))
        out = (ctypes.c_float * 4)()
        _check(lib().pseg_train_forward_backward_f32(self._h, _ptr(img), _ptr(m), img.shape[0], img.shape[1], out))
        return tuple(float(v) for v in out)

    def eval_step(self, image, mask):
        img, msk = self._img_mask(image, mask)
        m = (ctypes.c_float * 4)()
        _check(lib().pseg_eval_step(self._h, _ptr(img), _ptr(msk), img.shape[0], img.shape[1], m))
        return tuple(float(v) for v in m)

    def train_apply(self, lr, grad_scale=1.0):
        _check(lib().pseg_train_apply(self._h, float(lr), float(grad_scale)))

    def grad_buffer(self):
        """(device pointer, float count) of the flat gradient buffer -- what DP all-reduces."""
        ptr, n = ctypes.c_void_p(), ctypes.c_int64()
        _check(lib().pseg_train_grad_buffer(self._h, ctypes.byref(ptr), ctypes.byref(n)))
        return int(ptr.value), int(n.value)

    def gradients(self):
        L = lib()
        out = {}
        for name, shp in self.weight_specs():
            a = np.empty(shp, np.float32)
            _check(L.pseg_train_get_gradient(self._h, name.encode(), _ptr(a), a.size))
            out[name] = a
        return out

    def stream(self):
        """Raw hipStream_t (int) of the engine's own stream."""
        return int(lib().pseg_engine_stream(self._h) or 0)

    def flops_per_pixel(self):
        return float(lib().pseg_flops_per_pixel(self._h))

    # -- per-kernel timing (HIP events on the launch stream) -------------------------------------
    def timing_enable(self, on=True):
        _check(lib().pseg_timing_enable(self._h, int(bool(on))))

    def timing_reset(self):
        _check(lib().pseg_timing_reset(self._h))

    def timing(self):
        """[(layer, total_ms, launches, algorithmic_flops_per_launch)]"""
        L = lib()
        out = []
        name = ctypes.create_string_buffer(128)
        ms, n, fl = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
        for i in range(L.pseg_timing_num_slots(self._h)):
            _check(L.pseg_timing_get(self._h, i, name, 128, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl)))
            out.append((name.value.decode(), ms.value, n.value, fl.value))
        return out


# -- post-process (host arrays) ----------------------------------------------------------------
def cc_vote(pred, binary, n_classes=0, device=0):
    """vote_connected_component_class (lib/postprocess.py:9-26); `pred` int64 is updated IN PLACE
    (as the reference does) and returned."""
    if pred.dtype != np.int64 or not pred.flags.c_contiguous:
        raise PsegError("pred must be a C-contiguous int64 array")
    b = np.ascontiguousarray(binary, dtype=np.uint8)
    if b.shape != pred.shape:
        raise PsegError("binary shape %r != pred shape %r" % (b.shape, pred.shape))
    H, W = pred.shape
    _check(lib().pseg_cc_vote(int(device), _ptr(pred), _ptr(b), H, W, int(n_classes)))
    return pred


def bbox_fill(pred, n_classes=0, device=0):
    p = np.ascontiguousarray(pred, dtype=np.int64)
    out = np.zeros_like(p)
    H, W = p.shape
    _check(lib().pseg_bbox_fill(int(device), _ptr(p), _ptr(out), H, W, int(n_classes)))
    return out


def masks(pred, binary, lut, device=0):
    """generate_output_masks (lib/output.py:44-60) -> (color, overlay, inverted, fg_color)."""
    p = np.ascontiguousarray(pred, dtype=np.int64)
    b = np.ascontiguousarray(binary, dtype=np.uint8)
    t = np.ascontiguousarray(lut, dtype=np.uint8).reshape(-1, 3)
    H, W = p.shape
    outs = [np.empty((H, W, 3), np.uint8) for _ in range(4)]
    _check(lib().pseg_masks(int(device), _ptr(p), _ptr(b), _ptr(t), t.shape[0], H, W,
                            *[_ptr(o) for o in outs]))
    return tuple(outs)


def otsu_char_height(gray, inverse=False, device=0):
    """(char_height or None, otsu_threshold) -- lib/image_ops.py:58-82 without the file read."""
    g = np.ascontiguousarray(gray, dtype=np.uint8)
    H, W = g.shape
    h, t = ctypes.c_int(), ctypes.c_int()
    _check(lib().pseg_otsu_char_height(int(device), _ptr(g), H, W, int(bool(inverse)),
                                       ctypes.byref(h), ctypes.byref(t)))
    return (None if h.value < 0 else h.value), t.value


# ---- line-height normalisation (lib/dataset.py:114-150, lib/util.py:21-29) --------------------------

def rescale_shape(shape, scale):
    """np.round(scale * shape) of skimage.transform.rescale (half to even)."""
    ho, wo = ctypes.c_int(), ctypes.c_int()
    _check(lib().pseg_rescale_shape(int(shape[0]), int(shape[1]), float(scale), ctypes.byref(ho), ctypes.byref(wo)))
    return ho.value, wo.value


def aa_kernels(in_shape, out_shape):
    """The per-axis anti-aliasing kernels exactly as scipy.ndimage.gaussian_filter builds them with
    this process's NumPy (host set-up arithmetic: a dozen numbers per axis).  -> [(w or None, radius)] x 2."""
    out = []
    for n_in, n_out in zip(in_shape[:2], out_shape[:2]):
        sigma = max(0.0, (float(n_in) / float(n_out) - 1) / 2)
        if sigma <= 1e-15:
            out.append((None, 0))
            continue
        radius = int(4.0 * sigma + 0.5)
        x = np.arange(-radius, radius + 1)
        phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
        out.append((np.ascontiguousarray((phi / phi.sum())[::-1], dtype=np.float64), radius))
    return out


def _kptr(k):
    return _ptr(k[0]) if k[0] is not None else None


def resize_nearest(image, out_shape, device=0):
    """Order-0 resize that keeps values and dtype (gather on the GPU); (H,W) or (H,W,C) arrays."""
    a = np.ascontiguousarray(image)
    H, W = a.shape[:2]
    Ho, Wo = int(out_shape[0]), int(out_shape[1])
    eb = a.itemsize * int(np.prod(a.shape[2:], dtype=np.int64))
    if a.dtype == np.bool_:
        a = a.view(np.uint8)
    out = np.empty((Ho, Wo) + a.shape[2:], a.dtype)
    _check(lib().pseg_resize_nearest(int(device), _ptr(a), H, W, eb, _ptr(out), Ho, Wo))
    return out


def scale_image(image, out_shape, device=0):
    """scale_image (lib/dataset.py:122-128) -> float64 (Ho,Wo)."""
    a = np.asarray(image)
    f64 = a.dtype != np.uint8
    a = np.ascontiguousarray(a, dtype=np.float64 if f64 else np.uint8)
    H, W = a.shape
    Ho, Wo = int(out_shape[0]), int(out_shape[1])
    ky, kx = aa_kernels((H, W), (Ho, Wo))
    out = np.empty((Ho, Wo), np.float64)
    _check(lib().pseg_scale_image(int(device), _ptr(a), int(f64), H, W, _ptr(out), Ho, Wo,
                                  _kptr(ky), ky[1], _kptr(kx), kx[1]))
    return out


def prepare_images(image, binary, scale, max_width=None, device=0, want_stage1=False):
    """prepare_images (lib/dataset.py:131-150) -> (img uint8, bin uint8, orig_bin uint8[, stage1])."""
    img = np.ascontiguousarray(image, dtype=np.uint8)
    b = np.ascontiguousarray(binary, dtype=np.uint8)
    if img.ndim != 2 or img.shape != b.shape:
        raise PsegError("image and binary must be 2-D arrays of one shape, got %s and %s" % (img.shape, b.shape))
    H0, W0 = img.shape
    H1, W1 = rescale_shape((H0, W0), scale)
    H2 = W2 = 0
    if max_width is not None:
        n_scale = max_width / W1
        if n_scale < 1.0:
            H2, W2 = rescale_shape((H1, W1), n_scale)
    k1 = aa_kernels((H0, W0), (H1, W1))
    k2 = aa_kernels((H1, W1), (H2, W2)) if H2 else [(None, 0), (None, 0)]
    Hf, Wf = (H2, W2) if H2 else (H1, W1)
    o_img, o_bin = np.empty((Hf, Wf), np.uint8), np.empty((Hf, Wf), np.uint8)
    o_orig = np.empty((H0, W0), np.uint8)
    st1 = np.empty((H1, W1), np.float64) if want_stage1 else None
    _check(lib().pseg_prepare_images(int(device), _ptr(img), _ptr(b), H0, W0, H1, W1,
                                     _kptr(k1[0]), k1[0][1], _kptr(k1[1]), k1[1][1], H2, W2,
                                     _kptr(k2[0]), k2[0][1], _kptr(k2[1]), k2[1][1],
                                     _ptr(o_img), _ptr(o_bin), _ptr(o_orig), _ptr(st1) if st1 is not None else None))
    return (o_img, o_bin, o_orig, st1) if want_stage1 else (o_img, o_bin, o_orig)


FILL_MODES = {"nearest": 0, "constant": 1, "reflect": 2, "wrap": 3}


def affine_warp(plane, matrix, offset, order, device=0, fill_mode="nearest", cval=0.0):
    """scipy.ndimage.affine_transform(plane, matrix, offset, order=order, mode=fill_mode, cval=cval) for a float32 (H,W)
    plane, order 0 or 3, fill_mode 'nearest', 'constant', 'reflect' or 'wrap' (the four keras-preprocessing accepts), on the GPU
    (the augmentation warp of lib/data_generator.py)."""
    if fill_mode not in FILL_MODES:
        raise PsegError("affine_warp: unknown fill_mode %r" % (fill_mode,))
    a = np.ascontiguousarray(plane, dtype=np.float32)
    if a.ndim != 2:
        raise PsegError("affine_warp takes one (H,W) plane")
    m = np.ascontiguousarray(matrix, dtype=np.float64).reshape(4)
    o = np.ascontiguousarray(offset, dtype=np.float64).reshape(2)
    out = np.empty_like(a)
    _check(lib().pseg_affine_warp_fill(int(device), _ptr(a), a.shape[0], a.shape[1], _ptr(m), _ptr(o), int(order),
                                       FILL_MODES[fill_mode], float(cval), _ptr(out)))
    return out


def batch_units(shapes, cap=8):
    """The units pseg_predict_batch would cut a list of page shapes into: [(first page, page count), ...] (host logic, no GPU)."""
    n = len(shapes)
    H = (ctypes.c_int * max(n, 1))(*[int(s[0]) for s in shapes])
    W = (ctypes.c_int * max(n, 1))(*[int(s[1]) for s in shapes])
    first, count = (ctypes.c_int * max(n, 1))(), (ctypes.c_int * max(n, 1))()
    nu = lib().pseg_batch_units(n, H, W, int(cap), first, count, n)
    _check(min(nu, 0))
    return [(first[u], count[u]) for u in range(nu)]


def brightness_shift(x, brightness, device=0):
    """keras-preprocessing 1.1.2 apply_brightness_shift(x, brightness, scale=False) on a float array (H,W[,C]) -> float32,
    on the GPU (pseg_brightness_shift; lib/trainer.py:21 brightness_range)."""
    a = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(a)
    _check(lib().pseg_brightness_shift(int(device), _ptr(a), a.size, float(brightness), _ptr(out)))
    return out


def _labels_arg(a):
    """A label map as a contiguous array of 1-, 4- or 8-byte integers (no copy for uint8 / int32 / int64)."""
    a = np.asarray(a)
    if a.dtype == np.bool_:
        a = a.view(np.uint8)
    if a.dtype not in (np.uint8, np.int32, np.int64):
        a = a.astype(np.int64)
    return np.ascontiguousarray(a)


def eval_confusion(pred, mask, binary, n_classes, device=0):
    """counts[b][m][p] (2, n_classes+1, n_classes+1) int64: the joint histogram of (binary != 0, mask label,
    predicted label) behind fgpa / fgoverlap_per_class / count_matches / total_accuracy."""
    p, m = _labels_arg(pred), _labels_arg(mask)
    if p.shape != m.shape:
        raise PsegError("pred %r and mask %r differ in shape" % (p.shape, m.shape))
    b = None
    if binary is not None:
        b = np.ascontiguousarray(np.asarray(binary) != 0).view(np.uint8)
        if b.shape != p.shape:
            raise PsegError("binary %r and pred %r differ in shape" % (b.shape, p.shape))
    k = int(n_classes) + 1
    out = np.zeros((2, k, k), np.int64)
    _check(lib().pseg_eval_confusion(int(device), _ptr(p), p.dtype.itemsize, _ptr(m), m.dtype.itemsize,
                                     _ptr(b) if b is not None else None, ctypes.c_int64(p.size), int(n_classes), _ptr(out)))
    return out


def cc_label(binary, connectivity=4, device=0):
    """(num_labels, labels int32) as cv2.connectedComponents(binary, connectivity=...)."""
    b = np.ascontiguousarray(binary, dtype=np.uint8)
    if b.ndim != 2:
        raise PsegError("binary must be 2-dimensional")
    labels = np.zeros(b.shape, np.int32)
    n = ctypes.c_int32()
    _check(lib().pseg_cc_label(int(device), _ptr(b), b.shape[0], b.shape[1], int(connectivity), _ptr(labels), ctypes.byref(n)))
    return int(n.value), labels


def cc_tables(labels, num_labels, pred=None, mask=None, n_classes=0, want_stats=True, want_order=False, device=0):
    """dict with stats (N,5) int32 / centroids (N,2) float64 (cv2's), and with pred+mask: eq (N,), hist_pred,
    hist_mask (N, n_classes+1) int64; order (H*W,) int32 when asked."""
    lab = np.ascontiguousarray(labels, dtype=np.int32)
    n = int(num_labels)
    out = {}
    stats = cent = eq = hp = hm = order = None
    if want_stats:
        stats, cent = np.zeros((n, 5), np.int32), np.zeros((n, 2), np.float64)
        out.update(stats=stats, centroids=cent)
    p = m = None
    if pred is not None and mask is not None:
        p, m = _labels_arg(pred), _labels_arg(mask)
        if p.shape != lab.shape or m.shape != lab.shape:
            raise PsegError("labels, pred and mask differ in shape")
        k = int(n_classes) + 1
        eq, hp, hm = np.zeros(n, np.int64), np.zeros((n, k), np.int64), np.zeros((n, k), np.int64)
        out.update(eq=eq, hist_pred=hp, hist_mask=hm)
    if want_order:
        order = np.zeros(lab.size, np.int32)
        out["order"] = order
    opt = lambda a: _ptr(a) if a is not None else None
    _check(lib().pseg_cc_tables(int(device), _ptr(lab), lab.shape[0], lab.shape[1], n, opt(p), p.dtype.itemsize if p is not None else 0,
                                opt(m), m.dtype.itemsize if m is not None else 0, int(n_classes), opt(stats), opt(cent), opt(eq),
                                opt(hp), opt(hm), opt(order)))
    return out
