"""Objective functions that the fused step kernel can evaluate on the device.

The reference lets the user replace the pandapower cost tables by any Python callable
`objective_function(net) -> np.ndarray` (opf_env.py:52,80-84).  A Python callable cannot run
inside a kernel; the objectives here describe themselves, are compiled into
`opfx_env_desc.qterm_*`, and are ALSO plain callables with the reference's signature, so the
same object plugs into the reference's `OpfEnv` unchanged."""
import numpy as np


class QuadraticDeviation:
    """weight * (res_<unit>.<column> - target)^2 per element, e.g. the quadratic voltage deviation
    of examples/mixed_continuous_discrete.py:17-19: QuadraticDeviation('bus', 'vm_pu', 1.0)."""

    def __init__(self, unit='bus', column='vm_pu', target=1.0, weight=1.0, idxs=None):
        self.unit, self.column, self.target, self.weight, self.idxs = unit, column, float(target), float(weight), idxs

    def __call__(self, net) -> np.ndarray:           # exactly one parameter named `net` (opf_env.py:813-817)
        col = net['res_' + self.unit][self.column]
        if self.idxs is not None:
            col = col.loc[self.idxs]
        return self.weight * (np.asarray(col, dtype=float) - self.target) ** 2
