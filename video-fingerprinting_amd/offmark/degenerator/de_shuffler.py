"""Payload recovery.  Mirrors offmark.degenerator.de_shuffler.DeShuffler
(reference src/offmark/degenerator/de_shuffler.py:3-22): set_shape(payload_shape) -> self,
degenerate(wm) -> uint8 [L].

``degenerate_counts`` is the entry the GPU path uses: the detect kernel already produced
sum(bits[i::L]) per payload position, so only the mean / un-permute / threshold epilogue runs here."""
import numpy as np

from ..engine import payload_means


class DeShuffler:
    def __init__(self, key=None):
        self.key = key

    def set_shape(self, payload_shape):
        self.payload_shape = payload_shape
        self.payload_len = int(np.array(payload_shape).prod())
        self.payload_idx = np.arange(self.payload_len)
        np.random.RandomState(self.key).shuffle(self.payload_idx)
        return self

    def _finish(self, means: np.ndarray) -> np.ndarray:
        """means[..., i] = mean(bits[i::L]).  Undo the permutation, threshold at mid-range (strict >)."""
        payload = np.empty_like(means)
        payload[..., self.payload_idx] = means
        hi = payload.max(axis=-1, keepdims=True)
        lo = payload.min(axis=-1, keepdims=True)
        return (payload > 0.5 * (hi + lo)).astype(np.uint8)

    def degenerate(self, wm):
        bits = np.asarray(wm).flatten()
        L = self.payload_len
        with np.errstate(divide="ignore", invalid="ignore"):
            means = np.array([bits[i::L].mean() if i < bits.size else np.nan for i in range(L)], dtype=np.float64)
        return self._finish(means)

    def degenerate_counts(self, counts, n_bits: int):
        """counts: int array [..., L] of ones among bits[i::L]; n_bits: length of the bit vector (H*W//64)."""
        counts = np.asarray(counts)
        return self._finish(payload_means(counts, n_bits, self.payload_len))
