"""Predictor (reference: lib/predictor.py:10-54): page loop, optional rescale to the original
resolution, post-process chain, mask generation."""
import dataclasses
import os
from typing import Generator

import numpy as np

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

    # -- the chain on the device (pseg_predict_chain) ---------------------------------------------------------------
    def _chain_ops(self):
        """The post-processor list as chain op ids, or None when a foreign callable is in it (then the host chain runs)."""
        from . import postprocess as pp
        known = {pp.vote_connected_component_class: "cc_vote", pp.add_bounding_boxes: "bbox"}
        ops = []
        for processor in (self.settings.post_process or []):
            if processor not in known:
                return None
            ops.append(known[processor])
        return ops

    def _chain(self, data: SingleData, want_masks: bool):
        """predict -> [scale_to_original_shape] -> post-processors -> [masks] without the label map leaving the device
        (lib/predictor.py:32-54).  Returns (data', labels_u8, masks or None), or None when this page / these settings
        need the host chain (a post-processor that is not one of this package's, > 256 classes, no binary for the vote)."""
        net = self.network
        ops = self._chain_ops()
        if ops is None or net.n_classes > 256 or getattr(net, "_rgb", False) and np.asarray(data.image).ndim != 2:
            return None
        image = np.asarray(data.image)
        out_shape = None
        page = data
        binary = data.binary
        if self.settings.high_res_output:
            # lib/output.py:63-79: the image and the binarisation of the record are resized as the reference does (they are
            # not on the hot path); the label map's resize is a stage of the device chain (identity when the shapes agree)
            from .util import preserving_resize
            out_shape = tuple(int(v) for v in data.original_shape[:2])
            resized_image = preserving_resize(data.image, data.original_shape)
            if np.asarray(data.binary).shape != tuple(data.original_shape):
                binary = data.orig_binary if data.orig_binary is not None else preserving_resize(data.binary, data.original_shape).astype('bool')
            page = dataclasses.replace(data, binary=binary, image=resized_image)
        need_bin = want_masks or "cc_vote" in ops
        if need_bin and binary is None:
            return None
        from .util import gray_to_rgb
        img = gray_to_rgb(image) if getattr(net, "_rgb", False) else image
        res = net.model.predict_chain(img, binary=np.asarray(binary).astype(np.uint8) if need_bin else None, out_shape=out_shape,
                                      post_ops=ops, exact_labels=net.exact == "labels", labels=None if want_masks else "u8",
                                      lut=self.settings.color_map.lut() if want_masks else None, masks=want_masks)
        return page, res["labels"], res["masks"]

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
        """lib/predictor.py:27-30, a plain page loop in the reference.  With post-processors or high_res_output every
        page takes the device chain (predict_single); a bare label-map stream takes Network.predict_labels (the
        overlapped batch entry, BATCH_PAGES pages at a time).  Probabilities are fetched on first read."""
        pages = list(dataset.data)
        if (self.settings.post_process or self.settings.high_res_output) and self._chain_ops() is not None:
            for data in pages:
                yield self.predict_single(data)
            return
        for i in range(0, len(pages), self.BATCH_PAGES):
            chunk = pages[i:i + self.BATCH_PAGES]
            for data, pred in zip(chunk, self.network.predict_labels([d.image for d in chunk])):
                yield self._finish(data, pred)

    def predict_single(self, data: SingleData) -> Prediction:
        got = self._chain(data, want_masks=False)
        if got is not None:
            page, lab_u8, _ = got
            # labels: the reference's int64 map, widened on first read; probabilities: fetched on first read
            return Prediction(LazyArray(lambda lab_u8=lab_u8: lab_u8.astype(np.int64)),
                              LazyArray(lambda data=data: self.network.predict_single_data(data)[1]), page)
        data, prob, pred = self._labels(data)
        return Prediction(pred, prob, data)

    def predict_masks(self, data: SingleData) -> Masks:
        got = self._chain(data, want_masks=True)
        if got is not None:
            color, overlay, inverted, fg = got[2]
            return Masks(color=color, overlay=overlay, inverted_overlay=inverted, fg_color_mask=fg)
        data, _, pred = self._labels(data)
        return generate_output_masks(data, pred, self.settings.color_map)
