"""Predictor (reference: lib/predictor.py:10-54): page loop, optional rescale to the original
resolution, post-process chain, mask generation."""
import os
from typing import Generator

from .dataset import Dataset, SingleData
from .network import Network, tf_backend_allow_growth
from .output import Masks, generate_output_masks, scale_to_original_shape
from .predictor_data import LazyArray, Prediction, PredictSettings


class Predictor:
    def __init__(self, settings: PredictSettings, network: Network = None):
        self.settings = settings
        self.network = network
        if settings.gpu_allow_growth:
            tf_backend_allow_growth()
        if not network:
            self.network = Network("Predict", n_classes=settings.n_classes,
                                   model=os.path.abspath(self.settings.network))
        if settings.output:
            for sub in ("overlay", "color", "inverted"):
                os.makedirs(os.path.join(settings.output, sub), exist_ok=True)

    def _labels(self, data: SingleData):
        logit, prob, pred = self.network.predict_single_data(data)
        if self.settings.high_res_output:
            data, pred = scale_to_original_shape(data, pred)
        for processor in (self.settings.post_process or []):
            pred = processor(pred, data)
        return data, prob, pred

    #: pages per pseg_predict_batch call in predict(): uploads / downloads of neighbouring pages overlap the compute
    BATCH_PAGES = 8

    def _finish(self, data: SingleData, pred):
        """Everything of _labels() after the network: rescale to the original resolution, post-process chain."""
        page = data
        if self.settings.high_res_output:
            data, pred = scale_to_original_shape(data, pred)
        for processor in (self.settings.post_process or []):
            pred = processor(pred, data)
        prob = LazyArray(lambda page=page: self.network.predict_single_data(page)[1])
        return Prediction(pred, prob, data)

    def predict(self, dataset: Dataset) -> Generator[Prediction, None, None]:
        """lib/predictor.py:27-30, a plain page loop in the reference.  Here the label maps of BATCH_PAGES pages at a
        time come from Network.predict_labels (the overlapped batch entry); probabilities are fetched on first read."""
        pages = list(dataset.data)
        for i in range(0, len(pages), self.BATCH_PAGES):
            chunk = pages[i:i + self.BATCH_PAGES]
            for data, pred in zip(chunk, self.network.predict_labels([d.image for d in chunk])):
                yield self._finish(data, pred)

    def predict_single(self, data: SingleData) -> Prediction:
        data, prob, pred = self._labels(data)
        return Prediction(pred, prob, data)

    def predict_masks(self, data: SingleData) -> Masks:
        data, _, pred = self._labels(data)
        return generate_output_masks(data, pred, self.settings.color_map)
