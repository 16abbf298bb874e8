"""Throw-away stand-in for the `gymnasium` package, just enough for
`import opfgym` (the reference) to work inside this container when golden
vectors are generated (tests/golden/make_golden.py).  Not used by the product
and never shipped as a dependency."""
import numpy as np

from . import spaces  # noqa: F401
from . import envs  # noqa: F401


class RecordingRNG:
    """np.random.Generator look-alike that logs every uniform draw so that the
    batched environment can replay exactly the same random inputs.
    `uniform(low, high)` is `low + (high - low) * U[0,1)` as in numpy."""

    def __init__(self, seed=None):
        self._rng = np.random.default_rng(seed)
        self.log = []

    def random(self, size=None):
        u = self._rng.random(size)
        self.log.append(('random' if size is not None else 'random_scalar', np.array(u, copy=True)))
        return u

    def uniform(self, low=0.0, high=1.0, size=None):
        low, high = np.asarray(low, dtype=float), np.asarray(high, dtype=float)
        if size is None:
            size = np.broadcast(low, high).shape
        u = self._rng.random(size)
        self.log.append(('uniform', np.array(u, copy=True)))
        return low + (high - low) * u

    def choice(self, a, size=None):
        return self._rng.choice(a, size=size)

    def normal(self, loc=0.0, scale=1.0, size=None):
        # numpy: normal(loc, scale) = loc + scale * standard_normal()
        loc, scale = np.asarray(loc, dtype=float), np.asarray(scale, dtype=float)
        if size is None:
            size = np.broadcast(loc, scale).shape
        elif np.isscalar(size):
            size = (int(size),)
        z = self._rng.standard_normal(size)
        self.log.append(('normal', np.array(z, copy=True)))
        return loc + scale * z


class Env:
    def reset(self, seed=None, options=None):
        if seed is not None or not hasattr(self, 'np_random'):
            self.np_random = RecordingRNG(seed)

    @property
    def unwrapped(self):
        return self


class ObservationWrapper(Env):
    def __init__(self, env):
        self.env = env
