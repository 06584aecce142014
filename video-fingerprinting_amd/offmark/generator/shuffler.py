"""Bit-payload generator.  Mirrors offmark.generator.shuffler.Shuffler
(reference src/offmark/generator/shuffler.py:6-25): same constructor, wm_type() and
generate_wm(payload, capacity) -> int array of shape ``capacity``.  Host-side NumPy: the legacy
MT19937 ``RandomState(key).shuffle`` defines the permutation, so it stays on the CPU."""
import numpy as np


def tile_to_capacity(flat: np.ndarray, capacity) -> np.ndarray:
    total = int(np.prod(np.array(capacity)))
    reps = -(-total // flat.size)
    return np.tile(flat, reps)[:total].reshape(capacity)


class Shuffler:
    def __init__(self, key=None):
        self.key = key

    @staticmethod
    def wm_type():
        return "bits"

    def generate_wm(self, payload, capacity):
        """Permute a copy of the payload with the key, then repeat it up to ``capacity`` bits."""
        shuffled = np.copy(payload)
        np.random.RandomState(self.key).shuffle(shuffled)      # along axis 0, like the reference
        return tile_to_capacity(shuffled.reshape(-1), capacity)
