import numpy as np


class Box:
    def __init__(self, low, high, shape=None, seed=None, dtype=np.float32):
        if shape is None:
            shape = np.broadcast(np.asarray(low), np.asarray(high)).shape
        self.low = np.broadcast_to(np.asarray(low, dtype=float), shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=float), shape).copy()
        self.shape = tuple(shape)
        self._rng = np.random.default_rng(seed)

    def sample(self):
        return self.low + (self.high - self.low) * self._rng.random(self.shape)
