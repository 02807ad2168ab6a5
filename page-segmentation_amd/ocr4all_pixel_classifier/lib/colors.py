"""Minimal stand-in for ocr4all.colors.ColorMap (ocr4all-pylib 0.2.6 is not installed): only the
methods the hot path calls (lib/output.py:45, lib/dataset.py:181, lib/pagexml.py:124-129)."""
import json

import numpy as np


class ColorMap:
    def __init__(self, mapping=None):
        """mapping: {(r,g,b): (label_id, name)}"""
        self.mapping = {}
        for k, v in (mapping or {}).items():
            if isinstance(k, str):
                k = tuple(int(t) for t in k.strip("()[] ").split(","))
            self.mapping[tuple(int(t) for t in k)] = (int(v[0]), str(v[1]))

    @staticmethod
    def load(path):
        with open(path) as f:
            return ColorMap(json.load(f))

    def __len__(self):
        return len(self.mapping)

    def lut(self):
        """(n_labels, 3) uint8 table label -> RGB."""
        n = (max(v[0] for v in self.mapping.values()) + 1) if self.mapping else 1
        t = np.zeros((n, 3), np.uint8)
        for rgb, (lid, _) in self.mapping.items():
            t[lid] = rgb
        return t

    def to_rgb_array(self, labels):
        from pseg_amd import engine
        lab = np.ascontiguousarray(labels, dtype=np.int64)
        return engine.masks(lab, np.ones(lab.shape, np.uint8), self.lut())[0]

    def to_labels(self, rgb):
        """RGB (H,W,3) -> label ids; unknown colours -> 0."""
        rgb = np.asarray(rgb)[..., :3].astype(np.int64)
        key = (rgb[..., 0] << 16) | (rgb[..., 1] << 8) | rgb[..., 2]
        out = np.zeros(key.shape, np.uint8)
        for (r, g, b), (lid, _) in self.mapping.items():
            out[key == ((r << 16) | (g << 8) | b)] = lid
        return out

    def imread_labels(self, path):
        from PIL import Image
        return self.to_labels(np.asarray(Image.open(path).convert("RGB")))
