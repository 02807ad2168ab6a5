"""Network: the drop-in boundary (reference: lib/network.py).

Same constructor signature and method names as the reference's Network; the Keras model is
replaced by a pseg_amd.Engine handle (libpseg.so, HIP kernels on one MI355X).  Two keyword-only
extras select the GPU and the arithmetic mode.
"""
import logging
import os
from typing import Optional

import numpy as np

from pseg_amd import engine as _eng
from pseg_amd import synth as _synth

from .architecture import Architecture, Optimizers
from .callback import TrainProgressCallback
from .colors import ColorMap
from .dataset import Dataset, SingleData
from .util import gray_to_rgb

logger = logging.getLogger(__name__)


class Network:
    def __init__(self,
                 type: str,
                 n_classes: int = -1,
                 model_constructor: Architecture = Architecture.FCN_SKIP,
                 l_rate: float = 1e-4,
                 has_binary: bool = False,
                 foreground_masks: bool = False,
                 model: str = None,
                 continue_training: bool = False,
                 input_image_dimension: int = 1,
                 optimizer: Optimizers = Optimizers.ADAM,
                 optimizer_norm_clipping: bool = True,
                 optimizer_norm_clip_value: float = 1.0,
                 optimizer_clipping=False,
                 optimizer_clip_value=1,
                 loss_func=None,
                 *,
                 device: int = 0,
                 exact=None,
                 ):
        """
        :param type: "train" enables the training state, anything else ("Predict") is inference
        :param model: Keras HDF5 weight / full-model file (read by pseg_amd.h5lite, no h5py needed)
                      or the .npz written by save_weights(); no extension means '.h5' as in the reference
        :param device: HIP device index (keyword-only extension)
        :param exact: arithmetic of predict (keyword-only extension).  True (default, as the reference computes in
                      float32): float32 sequential-fmaf engine, bit-identical to the CPU oracle; 'labels': bf16 MFMA
                      throughput engine with the float32 referee on near-ties -- `pred` is the float32 engine's wherever
                      the referee looked and wherever the bf16 margin exceeds a threshold CALIBRATED (not proven) against
                      the float32 engine on every refereed crop; logits / probabilities carry bf16 accuracy.  exact=True
                      is the only mode that is bit-exact by construction.  False: bf16 throughput engine alone (label maps may
                      differ from float32 at near-ties of the two largest logits).  None reads PSEG_NETWORK_MODE
                      ('f32' default, 'bf16').
        """
        self.architecture = model_constructor.value
        self._data: Dataset = Dataset([], ColorMap({}))
        self.type = type
        self.has_binary = has_binary
        self.foreground_masks = foreground_masks
        self.l_rate = l_rate
        self.optimizer = optimizer
        self.optimizer_norm_clipping = optimizer_norm_clipping
        self.optimizer_norm_clip_value = optimizer_norm_clip_value
        self.optimizer_clipping = optimizer_clipping
        self.optimizer_clip_value = optimizer_clip_value
        self.loss_func = loss_func
        # lib/network.py:75-86: a saved full model decides the graph (tf.keras.models.load_model) -- its name replaces
        # the constructor's architecture (:251) and its logits kernel the class count (PredictSettings.n_classes
        # defaults to -1, lib/predictor_data.py:19-26); only when no file loads is model_constructor built (:89)
        path = self._resolve(model)
        file_w = None
        if path is not None and os.path.exists(path):
            file_w = self._read_weight_file(path)
            arch, n_cls, file_in_ch = self._graph_of_file(path, file_w)
            if arch is not None:
                self.architecture = arch
            if n_classes < 1:
                n_classes = n_cls
            if file_in_ch in (1, 3):
                input_image_dimension = file_in_ch
        if n_classes < 1:
            raise Exception("n_classes is not set and no model file provides it (model=%r)" % (model,))
        self.n_classes = n_classes
        _, rgb = Architecture(self.architecture).preprocess()
        self._rgb = rgb
        in_ch = 3 if rgb else input_image_dimension
        if exact is None:
            exact = os.environ.get("PSEG_NETWORK_MODE", "f32") != "bf16"
        self.exact = exact
        if exact is True:
            mode = _eng.MODE_F32_EXACT
        elif exact in (False, "labels"):
            mode = _eng.MODE_BF16
        else:
            raise Exception("exact must be True (float32), 'labels' (bf16 + float32 referee) or False (bf16)")
        self.model = _eng.Engine(Architecture(self.architecture).model(), n_classes, in_channels=in_ch, device=device, mode=mode)
        if file_w is not None:
            self._set_file_weights(path, file_w)
        else:
            if model and continue_training:
                raise Exception("Model file %s not found, cannot continue training" % model)
            # untrained graph, Keras-default glorot_uniform kernels and zero biases
            # (lib/network.py:89: model_constructor.model()(...) builds with default initialisers)
            seed = int(np.random.randint(0, 2 ** 31 - 1))
            self.model.set_weights(_synth.glorot_weights(self.model.weight_specs(), seed=seed))
            if model:
                logger.warning("model file %s not found: network starts from random weights", model)

    @staticmethod
    def _resolve(model):
        """lib/network.py:59: a model name without extension means <name>.h5.  A Keras HDF5 file is read
        directly (pseg_amd.h5lite); a sibling .npz (this package's native format) is used when the
        .h5 does not exist."""
        if not model:
            return None
        if '.' not in os.path.basename(model):
            model = model + '.h5'
        if model.endswith('.h5') and not os.path.exists(model):
            alt = model[:-3] + '.npz'
            if os.path.exists(alt):
                return alt
        return model

    # -- weights I/O (replaces ModelCheckpoint / load_weights, lib/network.py:106-107,177-183) --
    def save_weights(self, path):
        """'.h5' -> a Keras weights file (`model.load_weights` / this class read it back; layer and
        variable names are Keras' defaults: conv2d/kernel:0, ...); anything else -> .npz."""
        w = self.model.get_weights()
        if path.endswith('.h5'):
            from pseg_amd import h5lite
            layers = []
            for name, arr in w.items():
                lname = name.split('/')[0]
                if not layers or layers[-1][0] != lname:
                    layers.append((lname, []))
                layers[-1][1].append((name + ':0', arr))
            h5lite.write_keras_weights(path, self._keras_layer_order(layers))
            return path
        if not path.endswith('.npz'):
            path = os.path.splitext(path)[0] + '.npz'
        np.savez(path, **{k.replace('/', '__'): v for k, v in w.items()})
        return path

    @staticmethod
    def _read_weight_file(path):
        """-> [(weight name, array), ...] in the file's order."""
        if path.endswith('.h5'):
            from pseg_amd import h5lite
            return [(wn, a) for _, ws in h5lite.read_keras_weights(path) for wn, a in ws]
        with np.load(path, allow_pickle=False) as z:
            return [(k.replace('__', '/'), z[k]) for k in z.files]

    @staticmethod
    def _graph_of_file(path, file_w):
        """(architecture value | None, n_classes, input channels) of a model file: the saved model's name when the file
        carries one of the in-scope graphs' names (lib/model.py:91,202,233,306 name their models), else the weight-shape
        signature; classes = last dimension of the logits kernel (the last 4-D tensor)."""
        known = {a.value for a in (Architecture.FCN_SKIP, Architecture.FCN, Architecture.UNET, Architecture.RES_UNET)}
        name = None
        if path.endswith('.h5'):
            from pseg_amd import h5lite
            name = h5lite.read_keras_model_name(path)
        kernels = [a for _, a in file_w if a.ndim == 4]
        if not kernels:
            raise Exception("%s holds no convolution kernels" % path)
        n_cls = int(kernels[-1].shape[-1])
        first = kernels[0].shape
        arch = name if name in known else None
        if arch is None:
            if first[:2] == (5, 5) and first[3] == 20:
                # fcn vs fcn_skip: the third transposed conv reads 120 channels with skips, 60 without (lib/model.py:75,219)
                cin = [a.shape[3] for n, a in file_w if a.ndim == 4 and 'transpose' in n and a.shape[0] == 5]
                arch = Architecture.FCN_SKIP.value if (len(cin) > 1 and cin[1] == 120) else Architecture.FCN.value
            elif first[:2] == (3, 3) and first[3] == 64:
                arch = Architecture.UNET.value
            elif first[:2] == (3, 3) and first[3] == 32:
                arch = Architecture.RES_UNET.value
        return arch, n_cls, int(first[2])

    @staticmethod
    def _split_name(lname):
        base, _, suf = lname.rpartition('_')
        return (base, int(suf)) if base and suf.isdigit() else (lname, 0)

    def _keras_layer_order(self, layers):
        """model.layers order of the saved file for the engine's layer list (Keras creation order).  Keras sorts layers
        by depth and breaks ties by the order a traversal from the output meets them: in res_unet's residual_block
        (lib/model.py:243-249) the second conv and the shortcut conv tie and Add([shortcut, res]) reaches the shortcut
        first, so a file lists (conv_a, shortcut, conv_b) where creation order is (conv_a, conv_b, shortcut).  The
        other graphs are linear in depth."""
        if self.architecture != Architecture.RES_UNET.value:
            return list(layers)
        out = list(layers)
        blocks = [3 + 3 * i for i in range(4)] + [17 + 3 * i for i in range(4)]   # stem 0-2, encoder 3-14, bridge 15-16, decoder 17-28
        for b in blocks:
            out[b + 1], out[b + 2] = out[b + 2], out[b + 1]
        return out

    def load_weights(self, path):
        self._set_file_weights(path, self._read_weight_file(path))

    def _set_file_weights(self, path, file_w):
        specs = self.model.weight_specs()
        if not path.endswith('.h5'):
            self.model.set_weights(dict(file_w))
            return
        if len(file_w) != len(specs):
            raise Exception("%s holds %d weight tensors, the %s graph has %d"
                            % (path, len(file_w), self.architecture, len(specs)))
        # Names in a trained file carry Keras' per-process counters (conv2d_14, ...) and the file lists layers in
        # model.layers order, which is not creation order for res_unet: within each layer class (conv2d,
        # conv2d_transpose, logits) the numeric suffix IS creation order, the order of the engine's table.
        def by_class(names):
            groups = {}
            for idx, wn in enumerate(names):
                base, num = self._split_name(wn.split('/')[0])
                groups.setdefault(base, []).append((num, idx))
            return {b: [i for _, i in sorted(v)] for b, v in groups.items()}
        fg, sg = by_class([wn for wn, _ in file_w]), by_class([n for n, _ in specs])
        if {b: len(v) for b, v in fg.items()} == {b: len(v) for b, v in sg.items()}:
            pairs = [(si, fi) for base, sidx in sg.items() for si, fi in zip(sidx, fg[base])]
        else:
            # layer names that do not follow Keras' <class>_<n> pattern: file order, as load_weights(by_name=False)
            pairs = [(i, i) for i in range(len(specs))]
        out = {}
        for si, fi in pairs:
            (name, shape), (wn, a) = specs[si], file_w[fi]
            if tuple(a.shape) != tuple(shape):
                raise Exception("%s: %s has shape %s, %s expects %s" % (path, wn, a.shape, name, tuple(shape)))
            out[name] = np.ascontiguousarray(a, dtype=np.float32)
        self.model.set_weights(out)

    # -- predict (lib/network.py:248-260) --------------------------------------------------------
    def predict_single_data(self, data: SingleData):
        image = data.image
        if self._rgb:
            image = gray_to_rgb(image)
        if self.exact == "labels":
            # bf16 throughput pass for logits / probabilities, label map through the float32 referee (== float32 argmax)
            logit, prob, _ = self.model.predict(image, want_labels=False)
            pred = self.model.predict_exact_labels(image)
            return logit, prob, pred
        logit, prob, pred = self.model.predict(image)
        return logit, prob, pred

    def predict_labels(self, images):
        """int64 label maps (np.argmax of the logits, lib/network.py:259) of a list of pages without moving logits or
        probabilities off the device: the bf16 engine takes the overlapped batch entry (pseg_predict_batch), the
        'labels' mode the float32 referee per page, the float32 engine one predict per page."""
        imgs = [gray_to_rgb(im) if self._rgb else im for im in images]
        if self.exact is False and not self._rgb:
            return self.model.predict_batch(imgs, dtype=np.int64)
        if self.exact == "labels" and not self._rgb:
            return [self.model.predict_exact_labels(im) for im in imgs]
        return [self.model.predict(im, want_logits=False, want_probs=False)[2] for im in imgs]

    # -- training (lib/network.py:127-246) --------------------------------------------------------
    def create_dataset_inputs(self, train_data: Dataset, data_augmentation=True,
                              data_augmentation_settings=None, shuffle=False):
        """Infinite sample stream with the reference's structure (lib/network.py:127-165): in-place
        shuffle of the data list per pass when training, binary defaulting to ones, in-place
        foreground masking, batch of one; keys 'input_1' / 'input_2' / 'logits'."""
        from .architecture import default_preprocess
        from .util import image_to_batch
        augment = self.type == 'train' and data_augmentation
        if augment:
            if data_augmentation_settings is None:
                from .trainer import AugmentationSettings
                data_augmentation_settings = AugmentationSettings()
            image_gen, binary_gen, mask_gen = self._create_data_augmentation(data_augmentation_settings)
        data = train_data.data
        seed = 0
        while True:
            if self.type == 'train' and shuffle:
                np.random.shuffle(data)
            for d in data:
                b, i, m = d.binary, d.image, d.mask
                if self._rgb:
                    i = gray_to_rgb(i)
                if b is None:
                    b = np.full(i.shape, 1, dtype=np.uint8)
                    assert i.dtype == np.uint8
                if self.foreground_masks:
                    m[b != 1] = 0
                if augment:
                    seed += 1                                   # one seed per sample, shared by the three generators
                    i_n = next(image_gen.flow(image_to_batch(i), seed=seed, batch_size=1))
                    b_n = next(binary_gen.flow(image_to_batch(b), seed=seed, batch_size=1))
                    m_n = next(mask_gen.flow(image_to_batch(m), seed=seed, batch_size=1))
                    yield ({'input_1': default_preprocess(i_n), 'input_2': b_n}, {'logits': m_n})
                else:
                    yield ({'input_1': image_to_batch(default_preprocess(i)), 'input_2': image_to_batch(b)},
                           {'logits': image_to_batch(m)})

    def _create_data_augmentation(self, data_augmentation_settings):
        """lib/network.py:109-125: image (cubic), binary and mask (nearest) generators with shared settings."""
        from .data_generator import ImageDataGeneratorCustom
        image_gen = ImageDataGeneratorCustom(**data_augmentation_settings.to_image_params(), data_format='channels_last')
        binary_gen = ImageDataGeneratorCustom(**data_augmentation_settings.to_binary_params(), data_format='channels_last')
        mask_gen = ImageDataGeneratorCustom(**data_augmentation_settings.to_mask_params(), data_format='channels_last')
        return image_gen, binary_gen, mask_gen

    def _ensure_train_state(self):
        if getattr(self, "_train_ready", False):
            return
        if self.model.mode != _eng.MODE_F32_EXACT:
            raise Exception("training needs the float32 engine: construct Network(..., exact=True), the default")
        self.model.train_init(clipnorm=self.optimizer_norm_clip_value if self.optimizer_norm_clipping else 0.0,
                              clipvalue=self.optimizer_clip_value if self.optimizer_clipping else 0.0)
        self.model.train_set_optimizer(self.optimizer.value)      # Optimizers enum value = Keras name
        if self.loss_func is not None:
            self.model.train_set_loss(getattr(self.loss_func, "value", self.loss_func))   # Loss enum member or its value
        self._train_ready = True

    def _samples(self, dataset):
        for d in dataset.data:
            b, i, m = d.binary, d.image, d.mask
            if self.foreground_masks and b is not None:
                m[b != 1] = 0
            yield (gray_to_rgb(i) if self._rgb else i), m

    def evaluate_dataset(self, eval_data):
        """model.evaluate (lib/network.py:244-246): mean loss / accuracy / jacard / dice."""
        self._ensure_train_state()
        rows = [self.model.eval_step(i, m) for i, m in self._samples(eval_data)]
        mean = np.mean(np.asarray(rows, np.float64), axis=0) if rows else np.zeros(4)
        return dict(zip(("loss", "accuracy", "jacard_coef", "dice_coef"), (float(v) for v in mean)))

    def train_dataset(self, setting=None, callback: Optional[TrainProgressCallback] = None,
                      rank: int = 0, world: int = 1):
        """model.fit of lib/network.py:167-242 with the callbacks it configures: ModelCheckpoint
        (best only on `monitor`), EarlyStopping (patience = early_stopping_max_performance_drops,
        min_delta, restore best), ReduceLROnPlateau (factor, patience = drops / 2, min_lr) and the
        progress callback.  Batch of one page; with world > 1 every rank takes its own page per
        step and the gradients are averaged by one RCCL all-reduce (pseg_amd.parallel)."""
        from pseg_amd.parallel import dp_train_epoch, grad_tensor
        self._ensure_train_state()
        s = setting
        os.makedirs(s.output_dir, exist_ok=True)
        ckpt = os.path.join(s.output_dir, s.model_name + s.model_suffix)
        monitor = s.monitor.value
        maximise = ('acc' in monitor) or monitor.endswith('coef') or monitor.startswith('fmeasure')
        better = (lambda a, b: a > b + s.early_stopping_min_delta) if maximise else (lambda a, b: a < b - s.early_stopping_min_delta)
        best, wait, lr_wait = None, 0, 0
        best_weights = None
        lr = float(s.l_rate)
        history = {"loss": [], "accuracy": [], "jacard_coef": [], "dice_coef": [],
                   "val_loss": [], "val_accuracy": [], "lr": []}
        train = s.train_data.data
        n = len(train)
        if s.data_augmentation:
            image_gen, _, mask_gen = self._create_data_augmentation(s.data_augmentation_settings)
        it = 0
        for epoch in range(s.n_epoch):
            np.random.shuffle(train)                      # lib/network.py:134-135 (in place)
            def fb(k, epoch=epoch):
                d = train[k]
                img = gray_to_rgb(d.image) if self._rgb else d.image
                m = d.mask
                if self.foreground_masks and d.binary is not None:
                    m[d.binary != 1] = 0
                if not s.data_augmentation:
                    return self.model.train_forward_backward(img, m)
                # lib/network.py:149-161: one seed per sample (1, 2, ... in draw order; here epoch * n + k + 1,
                # the same numbers at world 1 and independent of the rank count), shared by the image (cubic)
                # and mask (nearest) warps; the float page keeps the 0..255 scale, the engine divides by 255
                seed = epoch * n + k + 1
                from .util import image_to_batch
                i_n = next(image_gen.flow(image_to_batch(img), seed=seed, batch_size=1))[0]
                m_n = next(mask_gen.flow(image_to_batch(m), seed=seed, batch_size=1))[0, ..., 0]
                if i_n.shape[-1] == 1:
                    i_n = i_n[..., 0]
                return self.model.train_forward_backward_float(i_n, m_n.astype(np.uint8))

            def apply(scale):
                nonlocal it
                self.model.train_apply(lr, scale)
                it += 1

            if world == 1:
                rows = []
                for k in range(n):
                    row = fb(k)
                    self.model.train_apply(lr, 1.0)
                    rows.append(row)
                    if callback:
                        callback.update_loss(it, row[0], row[1])
                    it += 1
            else:                                   # all ranks must share np.random's seed (same shuffle)
                rows = dp_train_epoch(n, rank, world, fb, lambda: grad_tensor(self.model), apply,
                                      engine_stream=self.model.stream())
                if callback:
                    for j, row in enumerate(rows):
                        callback.update_loss(it - len(rows) + j, row[0], row[1])
            logs = dict(zip(("loss", "accuracy", "jacard_coef", "dice_coef"),
                            (float(v) for v in np.mean(np.asarray(rows, np.float64), axis=0))))
            if s.validation_data is not None and len(s.validation_data) > 0:
                ev = self.evaluate_dataset(s.validation_data)
                logs.update({"val_" + k: v for k, v in ev.items()})
            for k in history:
                if k in logs:
                    history[k].append(logs[k])
            history["lr"].append(lr)
            cur = logs.get(monitor)
            if cur is None:
                logger.warning("monitor %s is not available (no validation data?), falling back to loss", monitor)
                cur, maximise_now = logs["loss"], False
                improved = best is None or cur < best - s.early_stopping_min_delta
            else:
                improved = best is None or better(cur, best)
            if improved:
                best, wait, lr_wait = cur, 0, 0
                if rank == 0 and (s.save_best_model_only or True):
                    self.save_weights(ckpt)
                if s.early_stopping_restore_best_weights:
                    best_weights = self.model.get_weights()
            else:
                wait += 1
                lr_wait += 1
                if rank == 0 and not s.save_best_model_only:
                    self.save_weights(ckpt)
                if s.reduce_lr_on_plateau and lr_wait >= s.early_stopping_max_performance_drops / 2:
                    lr = max(lr * s.reduce_lr_plateau_factor, s.reduce_lr_min_lr)
                    lr_wait = 0
            if callback:
                callback.next_best(it - 1, best, wait)
            if s.early_stopping_max_performance_drops != 0 and wait >= s.early_stopping_max_performance_drops:
                logger.info("early stopping after epoch %d", epoch + 1)
                break
        if best_weights is not None and s.early_stopping_restore_best_weights and wait > 0:
            self.model.set_weights(best_weights)
        self.history = history
        return history


def tf_backend_allow_growth():
    """lib/network.py:263-268 configures TensorFlow's allocator; nothing to do on this engine
    (device buffers are sized per page canvas and reused)."""
    return None
