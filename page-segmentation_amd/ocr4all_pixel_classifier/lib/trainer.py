"""TrainSettings / AugmentationSettings / Trainer (reference: lib/trainer.py).  The settings keep
the reference's field names, order and defaults (they are the API the frontend fills); training
runs on the float32 engine (Network(..., exact=True))."""
import logging
from typing import List, Optional

import numpy as np

from .architecture import Architecture, Optimizers
from .callback import TrainProgressCallback
from .dataset import Dataset
from .metrics import Loss, Monitor

logger = logging.getLogger(__name__)


def _record(name, fields):
    """A NamedTuple-compatible record type from a (field, type[, default]) table: the reference's field names,
    order and defaults (lib/trainer.py:13-29,59-106) are the API the frontend fills positionally and by keyword."""
    import collections
    T = collections.namedtuple(name, [f[0] for f in fields], defaults=[f[2] for f in fields if len(f) == 3])
    T.__annotations__ = {f[0]: f[1] for f in fields}
    T.__module__ = __name__
    return T


_AUGMENTATION_FIELDS = [
    ("rotation_range", float, 2.5), ("width_shift_range", float, 0.025), ("height_shift_range", float, 0.025),
    ("shear_range", float, 0.00), ("zoom_range", List[float], [0.95, 1.05]),
    ("horizontal_flip", bool, False), ("vertical_flip", bool, False), ("brightness_range", Optional[List[float]], None),
    ("image_fill_mode", str, 'nearest'), ("binary_fill_mode", str, 'nearest'), ("mask_fill_mode", str, 'nearest'),
    ("image_cval", int, 0), ("binary_cval", int, 0), ("mask_cval", int, 0),
]


class AugmentationSettings(_record("AugmentationSettings", _AUGMENTATION_FIELDS)):
    """Random affine augmentation of a training sample; the three generators (image: cubic, binary / mask: nearest)
    share everything but interpolation order, fill mode and fill value (lib/trainer.py:31-56)."""
    __slots__ = ()

    def _params(self, interp, fill_mode, cval, with_brightness):
        p = {k: getattr(self, k) for k in ("rotation_range", "width_shift_range", "height_shift_range", "shear_range",
                                          "zoom_range", "horizontal_flip", "vertical_flip")}
        p.update(interpolation_order=interp, fill_mode=fill_mode, cval=cval)
        if with_brightness:
            p['brightness_range'] = self.brightness_range
        return p

    def to_image_params(self):
        return self._params(3, self.image_fill_mode, self.image_cval, True)

    def to_binary_params(self):
        return self._params(0, self.binary_fill_mode, self.binary_cval, False)

    def to_mask_params(self):
        return self._params(0, self.mask_fill_mode, self.mask_cval, False)


TrainSettings = _record("TrainSettings", [
    # required, positional
    ("n_epoch", int), ("n_classes", int), ("l_rate", float), ("train_data", Dataset), ("validation_data", Dataset),
    ("display", int), ("output_dir", str), ("threads", int),
    # augmentation
    ("data_augmentation", bool, False), ("data_augmentation_settings", AugmentationSettings, AugmentationSettings()),
    # early stopping / learning-rate plateau (patience of the plateau = drops / 2, lib/network.py:222)
    ("early_stopping_max_performance_drops", int, 10), ("early_stopping_restore_best_weights", bool, True),
    ("early_stopping_min_delta", float, 0.0),
    ("reduce_lr_on_plateau", bool, True), ("reduce_lr_plateau_factor", float, 0.5), ("reduce_lr_min_lr", float, 0.000001),
    # checkpoint: <output_dir>/<model_name><model_suffix>
    ("model_name", str, 'model'), ("model_suffix", str, '.h5'), ("save_best_model_only", bool, True),
    ("save_weights_only", bool, False),
    # graph, loss, optimizer
    ("architecture", Architecture, Architecture.FCN_SKIP), ("loss", Loss, Loss.CATEGORICAL_CROSSENTROPY),
    ("monitor", Monitor, Monitor.VAL_LOSS), ("optimizer", Optimizers, Optimizers.ADAM),
    ("optimizer_norm_clipping", bool, True), ("optimizer_norm_clip_value", float, 1.0),
    ("optimizer_clipping", bool, False), ("optimizer_clip_value", float, 1.0),
    # the rest
    ("evaluation_data", Dataset, None), ("load", str, None), ("continue_training", bool, False),
    ("compute_baseline", bool, False), ("foreground_masks", bool, False), ("tensorboard", bool, False),
    ("image_dimension", int, 1), ("gpu_allow_growth", bool, False),
])


class Trainer:
    def __init__(self, settings: TrainSettings):
        self.settings = settings
        from .network import Network
        s = settings
        self.train_net = Network("train", s.n_classes, s.architecture, l_rate=s.l_rate,
                                 foreground_masks=s.foreground_masks, model=s.load,
                                 continue_training=s.continue_training,
                                 input_image_dimension=s.image_dimension, optimizer=s.optimizer,
                                 optimizer_norm_clipping=s.optimizer_norm_clipping,
                                 optimizer_norm_clip_value=s.optimizer_norm_clip_value,
                                 optimizer_clipping=s.optimizer_clipping,
                                 optimizer_clip_value=s.optimizer_clip_value, loss_func=s.loss,
                                 exact=True)
        if len(s.train_data) == 0 and s.n_epoch > 0:
            raise Exception("No training files specified. Maybe set n_iter=0")
        if s.compute_baseline:
            # lib/trainer.py:135-143: share of the most frequent label = trivial-classifier accuracy
            total = float(sum(d.mask.size for d in s.train_data.data))
            share = [sum(int(np.sum(d.mask == l)) for d in s.train_data.data) / total
                     for l in range(s.n_classes)]
            logging.info(f"Label percentage: {list(zip(range(s.n_classes), share))}")
            logging.info(f"Baseline: {max(share)}")

    def train(self, callback: Optional[TrainProgressCallback] = None) -> None:
        if callback:
            callback.init(self.settings.n_epoch * len(self.settings.train_data.data),
                          self.settings.early_stopping_max_performance_drops)
        return self.train_net.train_dataset(setting=self.settings, callback=callback)

    def eval(self) -> None:
        if self.settings.evaluation_data is None:
            logger.info('Evaluation Dataset in Trainsetting not set! ')
            return
        if len(self.settings.evaluation_data) > 0:
            return self.train_net.evaluate_dataset(self.settings.evaluation_data)
        else:
            logger.info('Empty Dataset. Skipping Evaluation')
