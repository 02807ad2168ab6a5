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
                 exact: bool = False,
                 ):
        """
        :param type: "train" enables the training state, anything else ("Predict") is inference
        :param model: weight file written by save_weights(); '.h5' paths are looked up as the
                      sibling '.npz' (Keras HDF5 needs h5py, which this image lacks)
        :param device: HIP device index (keyword-only extension)
        :param exact: True = float32 sequential-fmaf mode (bit-identical to the CPU oracle),
                      False = bf16 MFMA throughput mode (keyword-only extension)
        """
        self.architecture = model_constructor.value
        self._data: Dataset = Dataset([], ColorMap({}))
        self.type = type
        self.has_binary = has_binary
        self.foreground_masks = foreground_masks
        self.n_classes = n_classes
        self.l_rate = l_rate
        self.optimizer = optimizer
        self.optimizer_norm_clipping = optimizer_norm_clipping
        self.optimizer_norm_clip_value = optimizer_norm_clip_value
        self.optimizer_clipping = optimizer_clipping
        self.optimizer_clip_value = optimizer_clip_value
        self.loss_func = loss_func
        _, rgb = Architecture(self.architecture).preprocess()
        self._rgb = rgb
        in_ch = 3 if rgb else input_image_dimension

        self.model = _eng.Engine(model_constructor.model(), n_classes, in_channels=in_ch, device=device,
                                 mode=_eng.MODE_F32_EXACT if exact else _eng.MODE_BF16)
        path = self._resolve(model)
        if path is not None and os.path.exists(path):
            self.load_weights(path)
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
        if not model:
            return None
        if '.' not in os.path.basename(model):
            model = model + '.h5'                    # lib/network.py:59
        if model.endswith('.h5'):
            alt = model[:-3] + '.npz'
            if os.path.exists(alt):
                return alt
            if os.path.exists(model):
                raise Exception("Keras HDF5 model files need h5py, which is not available here; "
                                "convert %s to .npz (name -> array) first" % model)
            return alt
        return model

    # -- weights I/O (replaces ModelCheckpoint / load_weights, lib/network.py:106-107,177-183) --
    def save_weights(self, path):
        if not path.endswith('.npz'):
            path = os.path.splitext(path)[0] + '.npz'
        np.savez(path, **{k.replace('/', '__'): v for k, v in self.model.get_weights().items()})
        return path

    def load_weights(self, path):
        with np.load(path, allow_pickle=False) as z:
            self.model.set_weights({k.replace('__', '/'): z[k] for k in z.files})

    # -- predict (lib/network.py:248-260) --------------------------------------------------------
    def predict_single_data(self, data: SingleData):
        image = data.image
        if self._rgb:
            image = gray_to_rgb(image)
        logit, prob, pred = self.model.predict(image)
        return logit, prob, pred

    # -- training (lib/network.py:127-246): later SURVEY 8 row ------------------------------------
    def create_dataset_inputs(self, train_data: Dataset, data_augmentation=True,
                              data_augmentation_settings=None, shuffle=False):
        raise Exception("the training path (create_dataset_inputs / train_dataset) is not built yet "
                        "in the MI355X engine")

    def train_dataset(self, setting=None, callback: Optional[TrainProgressCallback] = None):
        raise Exception("the training path (train_dataset) is not built yet in the MI355X engine")

    def evaluate_dataset(self, eval_data):
        raise Exception("the training path (evaluate_dataset) is not built yet in the MI355X engine")


def tf_backend_allow_growth():
    """lib/network.py:263-268 configures TensorFlow's allocator; nothing to do on this engine
    (device buffers are sized per page canvas and reused)."""
    return None
