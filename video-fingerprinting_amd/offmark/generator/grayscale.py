"""Image-payload generator.  Mirrors offmark.generator.grayscale.GrayScale
(reference src/offmark/generator/grayscale.py:7-31)."""
import warnings

import numpy as np

from .shuffler import tile_to_capacity


class GrayScale:
    def __init__(self, key=None):
        self.key = key

    @staticmethod
    def wm_type():
        return "grayscale"

    def generate_wm(self, payload, capacity):
        """Threshold the image at 127, flatten, permute with the key, repeat up to ``capacity``."""
        total = int(np.prod(np.array(capacity)))
        if payload.size > total:
            warnings.warn(f"\nImage size {payload.shape} is greater than the embed's capacity: {total} pixels",
                          stacklevel=3)
        flat = (payload > 127).astype(np.uint8).flatten()
        np.random.RandomState(self.key).shuffle(flat)
        return tile_to_capacity(flat, capacity)
