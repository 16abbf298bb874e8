"""Native, parameterised problem definitions — the fallback of `definition.resolve` when the reference package
`opfgym` (with pandapower and simbench) is not importable.

What an OPF environment of the reference IS (`opfgym/envs/*.py: _define_opf` + the key lists of `__init__`,
`opfgym/simbench/build_simbench_net.py`) is expressed here as DATA: per environment class one `Recipe` —
a table of column rules `(table, column, expression)`, cost-row rules, and key rules — run by ONE small
interpreter over the element tables.  Expressions are built from `col(...)` (a column of the rule's own
table), `par(...)` (a constructor argument), `of(table, column)` (a column of another table, reduced to a
scalar) and constants with ordinary arithmetic / comparison operators, so that a constructor argument such
as `cos_phi`, `min_sgen_power` or `load_scaling` flows into the definition instead of selecting a file.

The recorded definitions under `opfgym_amd/definitions/` (written by `tests/golden/make_definitions.py` from the
reference's OWN classes) are regression fixtures for this module: `tests/test_native_definition.py` rebuilds every
recorded definition that has a recipe here and compares tables, keys and profile columns value for value.
"""
from __future__ import annotations

import operator
from dataclasses import dataclass, field

import numpy as np
import pandas as pd

from . import grids, net as ppn
from .definition import Definition


# ---------------------------------------------------------------------------------------------------
# expressions
# ---------------------------------------------------------------------------------------------------
class Expr:
    """Deferred value: `ev(frame, params, net)` -> scalar or Series aligned with `frame`."""

    def __init__(self, ev, text):
        self.ev, self.text = ev, text

    def __repr__(self):
        return self.text

    def _bin(self, other, op, sym, swap=False):
        o = lift(other)
        a, b = (o, self) if swap else (self, o)
        return Expr(lambda f, p, n: op(a.ev(f, p, n), b.ev(f, p, n)), f'({a.text} {sym} {b.text})')

    def __add__(self, o): return self._bin(o, operator.add, '+')
    def __radd__(self, o): return self._bin(o, operator.add, '+', True)
    def __sub__(self, o): return self._bin(o, operator.sub, '-')
    def __rsub__(self, o): return self._bin(o, operator.sub, '-', True)
    def __mul__(self, o): return self._bin(o, operator.mul, '*')
    def __rmul__(self, o): return self._bin(o, operator.mul, '*', True)
    def __truediv__(self, o): return self._bin(o, operator.truediv, '/')
    def __gt__(self, o): return self._bin(o, operator.gt, '>')
    def __neg__(self): return Expr(lambda f, p, n: -self.ev(f, p, n), f'-{self.text}')
    def __invert__(self): return Expr(lambda f, p, n: ~self.ev(f, p, n), f'~{self.text}')

    def abs(self):
        return Expr(lambda f, p, n: abs(self.ev(f, p, n)), f'|{self.text}|')


def lift(v) -> Expr:
    return v if isinstance(v, Expr) else Expr(lambda f, p, n: v, repr(v))


def col(name) -> Expr:
    return Expr(lambda f, p, n: f[name], name)


def par(name) -> Expr:
    return Expr(lambda f, p, n: p[name], f'${name}')


def of(table, column, reduce='max') -> Expr:
    return Expr(lambda f, p, n: getattr(n[table][column], reduce)(), f'{reduce}({table}.{column})')


def larger(a, b) -> Expr:
    a, b = lift(a), lift(b)
    return Expr(lambda f, p, n: np.maximum(a.ev(f, p, n), b.ev(f, p, n)), f'max({a.text}, {b.text})')


# ---------------------------------------------------------------------------------------------------
# recipe
# ---------------------------------------------------------------------------------------------------
@dataclass
class Poly:
    """One polynomial cost row per selected unit (pandapower `create_poly_cost`)."""
    et: str
    select: str                    # 'all' | 'controllable'
    cp1: object = 0.0
    cq2: object = None             # None: column left at its default 0


@dataclass
class Pwl:
    """One piece-wise linear cost row per selected unit (pandapower `create_pwl_cost`)."""
    et: str
    select: str
    points: list = field(default_factory=list)


@dataclass
class Recipe:
    defaults: dict                                     # constructor arguments and their defaults
    grid: dict = field(default_factory=dict)           # defaults of the grid preparation this class passes on
    first: list = field(default_factory=list)          # structural steps before the column rules: ('keep_first', table)
    columns: list = field(default_factory=list)        # (table, column, value | Expr)
    costs: list = field(default_factory=list)          # Poly / Pwl, in creation order
    cost_columns: list = field(default_factory=list)   # (table, column, value | Expr) on the cost tables afterwards
    require: list = field(default_factory=list)        # (description, predicate(net))
    obs: list = field(default_factory=list)            # (table, column, selector[, only-if parameter])
    act: list = field(default_factory=list)
    state: list = None                                 # None: same as obs
    n_minus_one: object = None


def _select(net, table, how):
    df = net[table]
    if how == 'all':
        return df.index
    flag = df['controllable'].to_numpy(bool)
    return df.index[flag if how == 'controllable' else ~flag]


def _assign(net, params, table, column, value):
    df = net[table]
    v = value.ev(df, params, net) if isinstance(value, Expr) else value
    df[column] = v


# ---------------------------------------------------------------------------------------------------
# grid preparation (what the reference's `build_simbench_net` does to a raw SimBench net + profiles)
# ---------------------------------------------------------------------------------------------------
GRID_DEFAULTS = dict(gen_scaling=1.0, load_scaling=1.0, storage_scaling=1.0, voltage_band=0.05, max_loading=80)

# which scaling argument a unit table follows
SCALING_OF = {'sgen': 'gen_scaling', 'gen': 'gen_scaling', 'load': 'load_scaling', 'storage': 'storage_scaling'}


def prepare_grid(net, profiles, gen_scaling=1.0, load_scaling=1.0, storage_scaling=1.0, voltage_band=0.05,
                 max_loading=80):
    """In place.  (1) `scaling` column per unit table; (2) voltage band on the buses, loading limit on lines and
    two-winding transformers; (3) profile repair: negative sgen feed-in clipped to zero, units whose time series
    is constant removed from table and profile; (4) data ranges of every profiled column as `max_max_* / min_min_*`
    (times scaling; symmetric about zero for storages) plus `mean_* / std_dev_*`, and the residual-load range of the
    external grids.  build_simbench_net.py:5-97."""
    given = dict(gen_scaling=gen_scaling, load_scaling=load_scaling, storage_scaling=storage_scaling)
    for table, arg in SCALING_OF.items():
        net[table]['scaling'] = given[arg]
    if voltage_band:
        net['bus']['max_vm_pu'], net['bus']['min_vm_pu'] = 1 + voltage_band, 1 - voltage_band
    if max_loading:
        for table in ('line', 'trafo'):
            net[table]['max_loading_percent'] = max_loading
    feed_in = profiles[('sgen', 'p_mw')]
    feed_in[feed_in < 0.0] = 0.0
    for (table, _), series in profiles.items():
        flat = series.max(axis=0) == series.min(axis=0)
        units = net[table]
        units.drop(units[flat].index, inplace=True)
        series.drop(columns=series.columns[flat], inplace=True)
    ranges_from_profiles(net, profiles)
    return net, profiles


def ranges_from_profiles(net, profiles):
    """Step (4) of `prepare_grid` on its own (build_simbench_net.py:64-97); needs the `scaling` columns."""
    for (table, column), series in profiles.items():
        units = net[table]
        top, bottom = series.max(axis=0), series.min(axis=0)
        if table == 'storage':
            top = np.maximum(top.abs(), bottom.abs())
            bottom = -top
        units[f'max_max_{column}'] = top * units.scaling
        units[f'min_min_{column}'] = bottom * units.scaling
        units[f'mean_{column}'] = series.mean(axis=0)
        units[f'std_dev_{column}'] = series.std(axis=0)
    residual = profiles[('load', 'p_mw')].sum(axis=1) - profiles[('sgen', 'p_mw')].sum(axis=1)
    reactive = profiles[('load', 'q_mvar')].sum(axis=1)
    for series, column in ((residual, 'p_mw'), (reactive, 'q_mvar')):
        net['ext_grid'][f'max_max_{column}'] = series.max()
        net['ext_grid'][f'min_min_{column}'] = series.min()
        net['ext_grid'][f'mean_{column}'] = series.mean()
    return net, profiles


# ---------------------------------------------------------------------------------------------------
# the recipes
# ---------------------------------------------------------------------------------------------------
def _reactive_range(table, apparent):
    """max_s_mva and the symmetric reactive data range of a unit table"""
    return [(table, 'max_s_mva', apparent),
            (table, 'max_max_q_mvar', col('max_s_mva')),
            (table, 'min_min_q_mvar', -col('max_s_mva'))]


_VOLTAGE_CONTROL = Recipe(
    defaults=dict(simbench_network_name='1-MV-semiurb--1-sw', load_scaling=1.5, gen_scaling=1.3, cos_phi=0.95,
                  max_q_exchange=0.5, min_sgen_power=0.5, min_storage_power=0.5, market_based=False),
    columns=[('load', 'controllable', False),
             ('sgen', 'controllable', col('max_max_p_mw') > par('min_sgen_power')),
             *_reactive_range('sgen', col('max_max_p_mw') / par('cos_phi')),
             ('storage', 'controllable', col('max_max_p_mw') > par('min_storage_power')),
             *_reactive_range('storage', col('max_max_p_mw').abs()),
             ('ext_grid', 'max_q_mvar', par('max_q_exchange')),
             ('ext_grid', 'min_q_mvar', -par('max_q_exchange'))],
    costs=[Poly('sgen', 'controllable', cp1=0.03, cq2=0), Poly('storage', 'controllable', cp1=-0.03, cq2=0),
           Poly('ext_grid', 'all', cp1=0.03, cq2=0)],
    cost_columns=[('poly_cost', 'min_cq2_eur_per_mvar2', 0), ('poly_cost', 'max_cq2_eur_per_mvar2', 0.03)],
    require=[('VoltageControl needs a grid without generators (gen table)', lambda net: len(net['gen']) == 0)],
    obs=[('sgen', 'p_mw', 'all'), ('storage', 'p_mw', 'all'), ('load', 'p_mw', 'all'), ('load', 'q_mvar', 'all'),
         ('poly_cost', 'cq2_eur_per_mvar2', 'all', 'market_based')],
    act=[('sgen', 'q_mvar', 'controllable'), ('storage', 'q_mvar', 'controllable')])

_Q_MARKET = Recipe(**{**vars(_VOLTAGE_CONTROL), 'defaults': dict(
    simbench_network_name='1-MV-rural--0-sw', gen_scaling=1.0, load_scaling=1.5, min_sgen_power=0.2, cos_phi=0.95,
    max_q_exchange=0.1, market_based=True, min_storage_power=0.5)})

_ECO_DISPATCH = Recipe(
    defaults=dict(simbench_network_name='1-HV-urban--0-sw', gen_scaling=1.0, load_scaling=1.5, max_price_eur_gwh=0.5,
                  min_power=0),
    columns=[('ext_grid', 'vm_pu', 1.0), ('gen', 'vm_pu', 1.0), ('load', 'controllable', False),
             ('ext_grid', 'min_p_mw', 0), ('ext_grid', 'max_p_mw', of('sgen', 'max_max_p_mw', 'max')),
             ('sgen', 'min_p_mw', 0), ('sgen', 'max_p_mw', col('max_max_p_mw')),
             ('gen', 'min_p_mw', 0), ('gen', 'max_p_mw', col('max_max_p_mw')),
             ('sgen', 'controllable', col('max_max_p_mw') > par('min_power')), ('sgen', 'min_min_p_mw', 0),
             ('gen', 'controllable', True),
             ('gen', 'max_q_mvar', 0.0), ('gen', 'min_q_mvar', 0.0), ('sgen', 'max_q_mvar', 0.0), ('sgen', 'min_q_mvar', 0.0)],
    costs=[Pwl('ext_grid', 'all', [[0, 10000, 1]]), Poly('sgen', 'controllable', cp1=0), Poly('gen', 'controllable', cp1=0)],
    cost_columns=[('poly_cost', 'min_cp1_eur_per_mw', 0), ('poly_cost', 'max_cp1_eur_per_mw', par('max_price_eur_gwh')),
                  ('pwl_cost', 'cp1_eur_per_mw', 0.0), ('pwl_cost', 'min_cp1_eur_per_mw', 0),
                  ('pwl_cost', 'max_cp1_eur_per_mw', par('max_price_eur_gwh'))],
    obs=[('load', 'p_mw', 'all'), ('load', 'q_mvar', 'all'), ('poly_cost', 'cp1_eur_per_mw', 'all'),
         ('pwl_cost', 'cp1_eur_per_mw', 'all'), ('sgen', 'p_mw', 'fixed'), ('storage', 'p_mw', 'all'),
         ('storage', 'q_mvar', 'all')],
    act=[('sgen', 'p_mw', 'controllable'), ('gen', 'p_mw', 'controllable')])

_MAX_RENEWABLE = Recipe(
    defaults=dict(simbench_network_name='1-HV-mixed--1-sw', gen_scaling=0.8, load_scaling=0.8, min_storage_power=10,
                  min_sgen_power=24),
    first=[('keep_first', 'ext_grid')],
    columns=[('trafo', 'max_loading_percent', 100), ('load', 'controllable', False), ('ext_grid', 'vm_pu', 1.0),
             ('storage', 'controllable', col('max_max_p_mw') > par('min_storage_power')),
             ('storage', 'q_mvar', 0.0), ('storage', 'max_q_mvar', 0.0), ('storage', 'min_q_mvar', 0.0),
             ('storage', 'max_p_mw', col('max_max_p_mw')), ('storage', 'min_p_mw', col('min_min_p_mw')),
             ('sgen', 'controllable', col('max_max_p_mw') > par('min_sgen_power')),
             ('sgen', 'min_p_mw', 0.0), ('sgen', 'q_mvar', 0.0), ('sgen', 'max_q_mvar', 0.0), ('sgen', 'min_q_mvar', 0.0)],
    costs=[Poly('sgen', 'all', cp1=-30 / 1000)],
    obs=[('sgen', 'max_p_mw', 'all'), ('load', 'p_mw', 'all'), ('load', 'q_mvar', 'all'), ('storage', 'p_mw', 'fixed')],
    state=[('sgen', 'p_mw', 'all'), ('load', 'p_mw', 'all'), ('load', 'q_mvar', 'all'), ('storage', 'p_mw', 'fixed')],
    act=[('sgen', 'p_mw', 'controllable'), ('storage', 'p_mw', 'controllable')])

_STORAGE_REACH = larger(col('min_min_p_mw').abs(), col('max_max_p_mw').abs())
_LOAD_SHEDDING = Recipe(
    defaults=dict(simbench_network_name='1-MV-comm--2-sw', gen_scaling=1.6, load_scaling=2.2, min_load_power=0.6,
                  min_storage_power=1.0, max_p_exchange=8.0, storage_efficiency=0.95),
    columns=[('load', 'controllable', col('max_max_p_mw') > par('min_load_power')),
             ('load', 'min_min_p_mw', 0), ('load', 'min_p_mw', 0),
             ('storage', 'max_p_mw', _STORAGE_REACH), ('storage', 'min_p_mw', -col('max_p_mw')),
             ('storage', 'min_min_p_mw', col('min_p_mw')), ('storage', 'max_max_p_mw', col('max_p_mw')),
             ('storage', 'controllable', col('max_max_p_mw') > par('min_storage_power')),
             ('sgen', 'controllable', False),
             ('ext_grid', 'max_p_mw', par('max_p_exchange')), ('ext_grid', 'min_p_mw', -np.inf)],
    costs=[Poly('load', 'controllable', cp1=0), Pwl('storage', 'controllable', [[-1000, 0, 1], [0, 1000, 1]])],
    cost_columns=[('poly_cost', 'min_cp1_eur_per_mw', -10), ('poly_cost', 'max_cp1_eur_per_mw', 0),
                  ('pwl_cost', 'cp1_eur_per_mw', 0), ('pwl_cost', 'min_cp1_eur_per_mw', 0),
                  ('pwl_cost', 'max_cp1_eur_per_mw', 2), ('ext_grid', 'vm_pu', 1.0)],
    obs=[('sgen', 'p_mw', 'all'), ('load', 'max_p_mw', 'all'), ('load', 'q_mvar', 'all'), ('storage', 'p_mw', 'fixed'),
         ('poly_cost', 'cp1_eur_per_mw', 'all'), ('pwl_cost', 'cp1_eur_per_mw', 'all')],
    state=[('sgen', 'p_mw', 'all'), ('load', 'p_mw', 'all'), ('load', 'q_mvar', 'all'), ('storage', 'p_mw', 'fixed')],
    act=[('load', 'p_mw', 'controllable'), ('storage', 'p_mw', 'controllable')])

_SECURITY_CONSTRAINED = Recipe(
    defaults=dict(simbench_network_name='1-HV-urban--0-sw'),
    columns=[('sgen', 'controllable', True), ('sgen', 'max_p_mw', col('max_max_p_mw')),
             ('sgen', 'min_p_mw', col('min_min_p_mw')), ('sgen', 'max_q_mvar', 0), ('sgen', 'min_q_mvar', 0),
             ('load', 'controllable', False), ('gen', 'controllable', False), ('storage', 'controllable', False)],
    costs=[Poly('ext_grid', 'all', cp1=0.01)],
    obs=[('load', 'p_mw', 'all'), ('load', 'q_mvar', 'all')],
    act=[('sgen', 'p_mw', 'all')],
    n_minus_one=(('line', 'in_service', np.array([1, 3, 7])),))

RECIPES = {
    'opfgym.envs.VoltageControl': _VOLTAGE_CONTROL,
    'opfgym.envs.QMarket': _Q_MARKET,
    'opfgym.envs.EcoDispatch': _ECO_DISPATCH,
    'opfgym.envs.MaxRenewable': _MAX_RENEWABLE,
    'opfgym.envs.LoadShedding': _LOAD_SHEDDING,
    'opfgym.examples.security_constrained.SecurityConstrained': _SECURITY_CONSTRAINED,
}


# ---------------------------------------------------------------------------------------------------
# interpreter
# ---------------------------------------------------------------------------------------------------
def has_recipe(ref_path: str) -> bool:
    return ref_path in RECIPES


def build(ref_path: str, class_kwargs: dict, grid_seed: int = 0, prepare=None) -> Definition:
    """Definition of the reference class `ref_path` for `class_kwargs` on the stand-in grid of that name
    (`grids.get_grid`; real SimBench data needs the `simbench` package, i.e. the reference route)."""
    recipe = RECIPES[ref_path]
    unknown = set(class_kwargs) - set(recipe.defaults) - set(GRID_DEFAULTS)
    if unknown:
        raise TypeError(f'{ref_path}: unknown definition argument(s) {sorted(unknown)}')
    params = {**recipe.defaults, **{k: v for k, v in class_kwargs.items() if k in recipe.defaults}}
    grid_args = {**GRID_DEFAULTS, **{k: params[k] for k in GRID_DEFAULTS if k in params},
                 **{k: v for k, v in class_kwargs.items() if k in GRID_DEFAULTS}}
    code = params['simbench_network_name']
    if code not in grids.GRIDS:
        raise ImportError(f'{code!r} is not one of the synthetic stand-in grids {sorted(grids.GRIDS)}; real SimBench '
                          f'grids need the `simbench` package (install opfgym: the definition is then read off the '
                          f'reference class)')
    net, profiles = grids.get_grid(code, int(grid_seed))
    prepare_grid(net, profiles, **grid_args)
    if prepare is not None:                  # stand-in helper between grid preparation and the class's own rules
        prepare(net, profiles)
    for step, table in recipe.first:
        if step == 'keep_first' and len(net[table]) > 1:
            net[table] = net[table].iloc[0:1]
    for table, column, value in recipe.columns:
        _assign(net, params, table, column, value)
    for rule in recipe.costs:
        for idx in _select(net, rule.et, rule.select):
            if isinstance(rule, Poly):
                extra = {} if rule.cq2 is None else {'cq2_eur_per_mvar2': rule.cq2}
                ppn.create_poly_cost(net, idx, rule.et, cp1_eur_per_mw=rule.cp1, **extra)
            else:
                ppn.create_pwl_cost(net, idx, rule.et, points=[list(p) for p in rule.points])
            ppn.finalize(net)
    for table, column, value in recipe.cost_columns:
        _assign(net, params, table, column, value)
    for text, holds in recipe.require:
        if not holds(net):
            raise AssertionError(text)

    def keys(rules):
        out = []
        for rule in rules:
            table, column, how = rule[:3]
            if len(rule) > 3 and not params[rule[3]]:
                continue
            out.append((table, column, np.asarray(_select(net, table, 'not' if how == 'fixed' else how))))
        return out
    obs = keys(recipe.obs)
    return Definition(class_name=ref_path.rsplit('.', 1)[1], net=net, act_keys=keys(recipe.act), obs_keys=obs,
                      state_keys=keys(recipe.state) if recipe.state is not None else list(obs), profiles=profiles,
                      n_minus_one_keys=tuple(recipe.n_minus_one or ()), meta={'native': True, 'params': params})
