"""Constraint specifications for the fused device evaluator.

Host-side mirror of `/root/reference/opfgym/constraints.py`: same class names,
constructor parameters and default discovery (`create_default_constraints`,
constraints.py:195-226).  The numbers are evaluated on the GPU inside
`opfx_step` (csrc/opfx.hip, "constraints" section); these objects only carry
the parameters and say which result column / boundary columns they bind.
Arbitrary Python `get_values` callables (constraints.py:44-45) cannot run inside
a kernel: such a constraint is evaluated on the host after the launch
(opfgym_amd/host_fallback.py, one call per instance — the documented fallback).
"""
from __future__ import annotations

import numpy as np
import pandas as pd


class ApparentPower:
    """`get_values` of a custom constraint that the fused evaluator understands: apparent power
    sqrt(P^2 + Q^2) of res_<unit> (examples/custom_constraint.py:9-11).  Also a plain callable with
    the reference's signature, so the same object works in the reference's `Constraint`."""

    def __init__(self, unit_type: str = 'sgen'):
        self.unit_type = unit_type

    def __call__(self, net):
        res = net['res_' + self.unit_type]
        return (res.p_mw ** 2 + res.q_mvar ** 2) ** 0.5


class Constraint:
    def __init__(self, unit_type: str, values_column: str, get_values=None, get_boundaries=None,
                 only_worst_case_violations: bool = False, autoscale_violation=True,
                 scale_bounded_values: bool = False, penalty_factor: float = 1.0,
                 penalty_power: float = 1.0, violation_count_penalty: float = 0.0):
        # constraints.py:62-65.  An `ApparentPower` value object runs in the kernel; any other Python value
        # callable makes this a HOST constraint (opfgym_amd/host_fallback.py: evaluated per instance after
        # the launch).  A boundary callable of a device constraint is evaluated ONCE, when the environment is
        # compiled (boundaries that change per reset belong into min_/max_ table columns); of a host
        # constraint, per call.
        self.get_values, self.get_boundaries_fn = get_values, get_boundaries
        self.unit_type = unit_type
        self.values_column = 's_mva' if isinstance(get_values, ApparentPower) else values_column
        self.only_worst_case_violations = only_worst_case_violations
        self.autoscale_violation = autoscale_violation
        self.scale_bounded_values = scale_bounded_values
        self.penalty_factor = penalty_factor
        self.penalty_power = penalty_power
        self.violation_count_penalty = violation_count_penalty

    def autoscale_factor(self, net) -> float:
        """constraints.py:82-83: `violation *= autoscale_violation` when truthy
        (a bool True multiplies by 1 — defect D8 is reproduced)."""
        a = self.autoscale_violation
        return float(a) if a else 0.0

    def boundaries(self, net):
        """constraints.py:93-108: (min, max) arrays aligned with net[unit_type]
        rows, NaN where the boundary column is absent; p/q bounds × scaling."""
        tbl = net[self.unit_type]
        if self.get_boundaries_fn is not None:
            b = self.get_boundaries_fn(net)
            return tuple(np.asarray(b[w], dtype=float) if w in b else np.full(len(tbl), np.nan) for w in ('min', 'max'))
        out = []
        for which in ('min', 'max'):
            col = f'{which}_{self.values_column}'
            if col in tbl:
                b = pd.to_numeric(tbl[col], errors='coerce').to_numpy(float)
                if self.scale_bounded_values or ('scaling' in tbl
                                                 and self.values_column in ('p_mw', 'q_mvar')):
                    b = b * tbl['scaling'].to_numpy(float)
            else:
                b = np.full(len(tbl), np.nan)
            out.append(b)
        return out[0], out[1]


class VoltageConstraint(Constraint):
    def __init__(self, autoscale_violation=True, **args):
        if autoscale_violation is True:
            autoscale_violation = 20                   # constraints.py:133-135
        super().__init__('bus', 'vm_pu', autoscale_violation=autoscale_violation, **args)


class LineOverloadConstraint(Constraint):
    def __init__(self, autoscale_violation=True, **args):
        if autoscale_violation is True:
            autoscale_violation = 1 / 30               # constraints.py:144-146
        super().__init__('line', 'loading_percent', autoscale_violation=autoscale_violation, **args)


class TrafoOverloadConstraint(Constraint):
    def __init__(self, autoscale_violation=True, **args):
        if autoscale_violation is True:
            autoscale_violation = 1 / 30               # constraints.py:155-157
        super().__init__('trafo', 'loading_percent', autoscale_violation=autoscale_violation, **args)


class Trafo3wOverloadConstraint(Constraint):
    def __init__(self, autoscale_violation=True, **args):
        if autoscale_violation is True:
            autoscale_violation = 1 / 30               # constraints.py:164-172
        super().__init__('trafo3w', 'loading_percent', autoscale_violation=autoscale_violation, **args)


class ExtGridActivePowerConstraint(Constraint):
    def __init__(self, **args):
        super().__init__('ext_grid', 'p_mw', **args)

    def autoscale_factor(self, net) -> float:
        if not self.autoscale_violation:               # constraints.py:179-182
            self.autoscale_violation = 1 / abs(net.ext_grid['mean_p_mw'].sum())
        return float(self.autoscale_violation)


class ExtGridReactivePowerConstraint(Constraint):
    def __init__(self, **args):
        super().__init__('ext_grid', 'q_mvar', **args)

    def autoscale_factor(self, net) -> float:
        if not self.autoscale_violation:               # constraints.py:189-192
            self.autoscale_violation = 1 / abs(net.ext_grid['mean_q_mvar'].sum())
        return float(self.autoscale_violation)


def _defined(net, unit_type, column) -> bool:
    # constraints.py:229-238
    if unit_type not in net or column not in net[unit_type]:
        return False
    return bool(np.isfinite(pd.to_numeric(net[unit_type][column], errors='coerce')).any())


def create_default_constraints(net, constraint_kwargs: dict) -> list:
    """constraints.py:195-226."""
    out = []
    if _defined(net, 'bus', 'max_vm_pu') or _defined(net, 'bus', 'min_vm_pu'):
        out.append(VoltageConstraint(**constraint_kwargs))
    if _defined(net, 'line', 'max_loading_percent'):
        out.append(LineOverloadConstraint(**constraint_kwargs))
    if _defined(net, 'trafo', 'max_loading_percent'):
        out.append(TrafoOverloadConstraint(**constraint_kwargs))
    if _defined(net, 'trafo3w', 'max_loading_percent'):
        out.append(Trafo3wOverloadConstraint(**constraint_kwargs))
    if _defined(net, 'ext_grid', 'max_p_mw') or _defined(net, 'ext_grid', 'min_p_mw'):
        out.append(ExtGridActivePowerConstraint(**constraint_kwargs))
    if _defined(net, 'ext_grid', 'max_q_mvar') or _defined(net, 'ext_grid', 'min_q_mvar'):
        out.append(ExtGridReactivePowerConstraint(**constraint_kwargs))
    return out
