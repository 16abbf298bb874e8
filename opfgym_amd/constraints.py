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


# The reference's ready-made constraints (constraints.py:131-192) differ in four facts only, so they are rows of a
# table here: name, table, bounded result column, what `autoscale_violation=True` stands for — a fixed factor
# (:133-135, :144-146, :155-157, :164-172) or, for the slack exchange, nothing: `True` multiplies by one (defect D8) and a
# falsy value turns into 1 / |sum of the mean exchange| (:179-182, :189-192) — and the limit columns whose presence
# makes `create_default_constraints` add the constraint (:195-226).
_READY_MADE = (
    ('VoltageConstraint', 'bus', 'vm_pu', 20, None, ('max_vm_pu', 'min_vm_pu')),
    ('LineOverloadConstraint', 'line', 'loading_percent', 1 / 30, None, ('max_loading_percent',)),
    ('TrafoOverloadConstraint', 'trafo', 'loading_percent', 1 / 30, None, ('max_loading_percent',)),
    ('Trafo3wOverloadConstraint', 'trafo3w', 'loading_percent', 1 / 30, None, ('max_loading_percent',)),
    ('ExtGridActivePowerConstraint', 'ext_grid', 'p_mw', None, 'mean_p_mw', ('max_p_mw', 'min_p_mw')),
    ('ExtGridReactivePowerConstraint', 'ext_grid', 'q_mvar', None, 'mean_q_mvar', ('max_q_mvar', 'min_q_mvar')),
)


def _ready_made(name, table, column, factor_for_true, mean_column, _limits):
    def __init__(self, autoscale_violation=True, **args):
        if factor_for_true is not None and autoscale_violation is True:
            autoscale_violation = factor_for_true
        Constraint.__init__(self, table, column, autoscale_violation=autoscale_violation, **args)

    def autoscale_factor(self, net) -> float:
        if mean_column is not None and not self.autoscale_violation:
            self.autoscale_violation = 1 / abs(net[table][mean_column].sum())
        return Constraint.autoscale_factor(self, net)
    return type(name, (Constraint,), {'__init__': __init__, 'autoscale_factor': autoscale_factor,
                                      '__doc__': f'`opfgym.constraints.{name}`: {table}.{column} within its min_/max_ columns.'})


for _row in _READY_MADE:
    globals()[_row[0]] = _ready_made(*_row)


def _has_numbers(net, table, column) -> bool:
    """A limit column counts when it exists and holds at least one finite number (constraints.py:229-238)."""
    if table not in net or column not in net[table]:
        return False
    return bool(np.isfinite(pd.to_numeric(net[table][column], errors='coerce')).any())


def create_default_constraints(net, constraint_kwargs: dict) -> list:
    """The ready-made constraints whose limit columns the net carries, in the reference's order (constraints.py:195-226)."""
    return [globals()[name](**constraint_kwargs) for name, table, _, _, _, limits in _READY_MADE
            if any(_has_numbers(net, table, col) for col in limits)]
