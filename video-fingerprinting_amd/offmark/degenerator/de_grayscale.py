"""Image-payload recovery.  Mirrors offmark.degenerator.de_grayscale.DeGrayScale
(reference src/offmark/degenerator/de_grayscale.py:3-23): result is 0/255 in the payload's shape."""
import numpy as np

from .de_shuffler import DeShuffler


class DeGrayScale(DeShuffler):
    def degenerate(self, wm_bits):
        return (super().degenerate(wm_bits) * 255).astype(np.uint8).reshape(self.payload_shape)

    def degenerate_counts(self, counts, n_bits: int):
        out = super().degenerate_counts(counts, n_bits) * 255
        return out.astype(np.uint8).reshape(tuple(np.shape(counts)[:-1]) + tuple(np.atleast_1d(self.payload_shape)))
