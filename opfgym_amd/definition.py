"""Problem definitions come from the reference, not from this package.

What an OPF environment IS — which units are controllable, their limit columns, the cost tables, the
action / observation / state keys — is written down once, in the reference's environment classes
(`opfgym/envs/*.py`, `opfgym/examples/*.py`: `_define_opf` and the key lists of `__init__`).  This package does
not restate that code line by line.  A batched environment gets its definition in one of three ways:

  * from a LIVE reference environment: `BatchedOpfEnv.from_reference(ref_env, ...)` / `extract(ref_env)` read
    the constructed object (`ref_env.net`, `.act_keys`, `.obs_keys`, `.state_keys`, `.profiles`, ...);
    `resolve()` builds that object itself when `opfgym` (with pandapower and simbench) is importable;
  * from the NATIVE builder (`opfgym_amd/native_definition.py`): the benchmark classes as rule tables run by one
    interpreter, parameterised by the constructor arguments, on the synthetic stand-in grids — the fallback when
    `opfgym` is not importable (the GPU box);
  * from a recorded definition: a small `.npz` holding exactly that data (element tables, key lists, the
    surviving profile columns), written by `tests/golden/make_definitions.py` from the reference's own classes.
    The files under `opfgym_amd/definitions/` are (a) the regression fixtures the native builder is tested against
    and (b) the definitions of the reference's EXAMPLE classes, which have no native recipe.

Only what genuinely has to be re-expressed for the device stays in `envs.py`: the per-reset `_sampling` tails
as vector ops of the reset kernel.
"""
from __future__ import annotations

import importlib
import json
import os
from dataclasses import dataclass, field

import numpy as np
import pandas as pd

TABLES = ('bus', 'line', 'trafo', 'trafo3w', 'load', 'sgen', 'storage', 'gen', 'ext_grid', 'shunt', 'switch',
          'poly_cost', 'pwl_cost')
_INT_COLS = ('bus', 'from_bus', 'to_bus', 'hv_bus', 'mv_bus', 'lv_bus', 'element')
HERE = os.path.dirname(os.path.abspath(__file__))
DEF_DIR = os.path.join(HERE, 'definitions')
# True: classes with a native recipe are read from their recorded definition as well (a test / comparison switch, set by the
# caller — this module does not consult the process environment)
PREFER_RECORDED = False


@dataclass
class Definition:
    class_name: str
    net: object                          # attribute-dict of DataFrames (opfgym_amd.net.Net or a pandapowerNet)
    act_keys: list
    obs_keys: list
    state_keys: list
    profiles: object = None              # dict[(unit, column)] -> DataFrame, or None (no time series)
    n_minus_one_keys: tuple = ()
    meta: dict = field(default_factory=dict)


# ---------------------------------------------------------------------------------------------------
# tables <-> plain arrays
# ---------------------------------------------------------------------------------------------------
def tables_to_arrays(net) -> dict:
    """Element tables as plain arrays: numeric columns float64, flags int8, everything else strings
    (lists, e.g. `pwl_cost.points`, as JSON)."""
    out = {'scalar__sn_mva': np.array(float(net['sn_mva'])), 'scalar__f_hz': np.array(float(net['f_hz']))}
    for tbl in TABLES:
        if tbl not in net:
            continue
        df = net[tbl]
        out[f'idx__{tbl}'] = np.asarray(df.index, dtype=np.int64)
        for col in df.columns:
            vals = df[col].to_numpy()
            if vals.dtype == bool:
                out[f'tab__{tbl}__{col}'] = vals.astype(np.int8)
                continue
            try:
                out[f'tab__{tbl}__{col}'] = vals.astype(np.float64)
            except (TypeError, ValueError):
                if any(isinstance(v, (list, tuple, np.ndarray)) for v in vals):
                    out[f'json__{tbl}__{col}'] = np.array([json.dumps(np.asarray(v, dtype=float).tolist()) for v in vals])
                else:
                    out[f'str__{tbl}__{col}'] = np.array(['' if v is None or v != v else str(v) for v in vals])
    return out


def arrays_to_net(z):
    """Inverse of tables_to_arrays: an `opfgym_amd.net.Net`."""
    from .net import Net
    keys = z.files if hasattr(z, 'files') else list(z)
    net = Net('definition', f_hz=float(z['scalar__f_hz']), sn_mva=float(z['scalar__sn_mva']))
    cols = {}
    for key in keys:
        kind, _, rest = key.partition('__')
        if kind not in ('tab', 'str', 'json'):
            continue
        tbl, _, col = rest.partition('__')
        arr = np.asarray(z[key])
        if kind == 'str':
            arr = np.array([None if v == '' else str(v) for v in arr], dtype=object)
        elif kind == 'json':
            vals = [json.loads(str(v)) for v in arr]
            arr = np.empty(len(vals), dtype=object)
            for i, v in enumerate(vals):
                arr[i] = v
        elif arr.dtype == np.int8:
            arr = arr.astype(bool)
        cols.setdefault(tbl, {})[col] = arr
    for tbl in TABLES:
        if f'idx__{tbl}' not in keys:
            continue
        data = cols.get(tbl, {})
        idx = np.asarray(z[f'idx__{tbl}'])
        df = pd.DataFrame({c: pd.Series(v, index=idx) for c, v in data.items()}, index=idx)
        for col in _INT_COLS:
            if col in df.columns and len(df) and not df[col].isna().any():
                df[col] = df[col].astype(np.int64)
        net[tbl] = df
    return net


# ---------------------------------------------------------------------------------------------------
# live reference environment -> Definition
# ---------------------------------------------------------------------------------------------------
def _keys(keys):
    return [(str(u), str(c), np.asarray(list(i))) for u, c, i in keys]


def extract(ref_env) -> Definition:
    """Read the problem definition off a constructed reference environment (`opfgym.OpfEnv` subclass)."""
    net = ref_env.net
    n1 = tuple((u, c, np.asarray(list(i))) for u, c, i in getattr(ref_env, 'n_minus_one_keys', ()) or ())
    return Definition(class_name=type(ref_env).__name__, net=net, act_keys=_keys(ref_env.act_keys),
                      obs_keys=_keys(ref_env.obs_keys), state_keys=_keys(ref_env.state_keys),
                      profiles=getattr(ref_env, 'profiles', None), n_minus_one_keys=n1)


# ---------------------------------------------------------------------------------------------------
# recorded definitions
# ---------------------------------------------------------------------------------------------------
def canonical_key(ref_path: str, class_kwargs: dict) -> str:
    def norm(v):
        if isinstance(v, (np.ndarray, list, tuple, range)):
            return [norm(x) for x in v]
        if isinstance(v, (bool, np.bool_)):
            return bool(v)
        if isinstance(v, (int, np.integer)):
            return int(v)
        if isinstance(v, (float, np.floating)):
            return float(v)
        return v
    return json.dumps([ref_path, sorted((k, norm(v)) for k, v in class_kwargs.items())], sort_keys=True)


def save(defn: Definition, path: str, grid_code=None, grid_seed=0, raw_profiles=None, prepare=None) -> None:
    """Write a definition.  Time series are not stored: the file names the synthetic grid whose generator
    reproduces them (`grids.get_grid(grid_code, grid_seed)`) and, per profile table, the columns that survived
    the reference's profile repair (and a lower clip where the repair changed values)."""
    data = tables_to_arrays(defn.net)
    data['meta__class_name'] = np.array(defn.class_name)
    for kind, keys in (('act', defn.act_keys), ('obs', defn.obs_keys), ('state', defn.state_keys),
                       ('nm1', defn.n_minus_one_keys)):
        for k, (u, c, idx) in enumerate(keys):
            data[f'key__{kind}__{k}__{u}__{c}'] = np.asarray(idx, dtype=np.int64)
    if defn.profiles:
        data['meta__grid_code'], data['meta__grid_seed'] = np.array(str(grid_code)), np.array(int(grid_seed))
        if prepare:
            data['meta__prepare'] = np.array(str(prepare))
        for (u, c), df in defn.profiles.items():
            data[f'prof__{u}__{c}'] = np.asarray(df.columns, dtype=np.int64)
            if raw_profiles is not None:
                raw = raw_profiles[(u, c)][list(df.columns)].to_numpy()
                if not np.array_equal(raw, df.to_numpy()):
                    lo = float(df.to_numpy().min()) if df.size else 0.0
                    assert np.array_equal(np.maximum(raw, lo), df.to_numpy()), 'profile repair is not a lower clip'
                    data[f'clip__{u}__{c}'] = np.array(lo)
    np.savez_compressed(path, **data)


def load(path: str) -> Definition:
    from . import grids
    z = np.load(path, allow_pickle=False)
    net = arrays_to_net(z)
    keys = {'act': {}, 'obs': {}, 'state': {}, 'nm1': {}}
    for name in z.files:
        if name.startswith('key__'):
            _, kind, k, u, c = name.split('__', 4)
            keys[kind][int(k)] = (u, c, np.asarray(z[name]))
    lists = {kind: [v for _, v in sorted(d.items())] for kind, d in keys.items()}
    profiles = None
    if 'meta__grid_code' in z.files:
        raw_net, profiles = grids.get_grid(str(z['meta__grid_code']), int(z['meta__grid_seed']))
        if 'meta__prepare' in z.files:                    # stand-in helper applied before the reference class saw the grid
            from . import simbench_build
            getattr(simbench_build, str(z['meta__prepare']))(raw_net, profiles)
        for key in list(profiles):
            name = f'prof__{key[0]}__{key[1]}'
            if name not in z.files:
                del profiles[key]
                continue
            df = profiles[key][[int(v) for v in z[name]]]
            if f'clip__{key[0]}__{key[1]}' in z.files:
                df = df.clip(lower=float(z[f'clip__{key[0]}__{key[1]}']))
            profiles[key] = df
    return Definition(class_name=str(z['meta__class_name']), net=net, act_keys=lists['act'], obs_keys=lists['obs'],
                      state_keys=lists['state'], profiles=profiles, n_minus_one_keys=tuple(lists['nm1']),
                      meta={'path': path})


def _index():
    path = os.path.join(DEF_DIR, 'index.json')
    return json.load(open(path)) if os.path.exists(path) else {}


def reference_class(ref_path: str):
    """The reference's environment class, or None when `opfgym` (or one of its dependencies: gymnasium,
    pandapower, simbench) cannot be imported."""
    module, _, name = ref_path.rpartition('.')
    try:
        return getattr(importlib.import_module(module), name)
    except Exception:
        return None


def request_key(ref_path: str, class_kwargs: dict, grid_seed=0, prepare=None) -> str:
    extra = {}
    if grid_seed:
        extra['__grid_seed'] = int(grid_seed)
    if prepare:
        extra['__prepare'] = str(prepare)
    return canonical_key(ref_path, {**class_kwargs, **extra})


def build_from_reference(cls, class_kwargs: dict, prepare=None) -> Definition:
    """Construct the reference class and read its definition.  `prepare(net, profiles)`: applied to what the
    class's module-level `build_simbench_net` returns, before the class works on it."""
    module = importlib.import_module(cls.__module__)
    original = getattr(module, 'build_simbench_net', None)
    if prepare is not None and original is not None:
        def prepared(*a, **k):
            net, profiles = original(*a, **k)
            prepare(net, profiles)
            return net, profiles
        module.build_simbench_net = prepared
    try:
        return extract(cls(**class_kwargs))
    finally:
        if prepare is not None and original is not None:
            module.build_simbench_net = original


def resolve(ref_path: str, class_kwargs: dict, grid_seed=0, prepare=None) -> Definition:
    """Definition of the reference class `ref_path` constructed with `class_kwargs`: from the live reference
    when it is importable, else from the recorded file for exactly these arguments."""
    cls = reference_class(ref_path)
    if cls is not None:
        from . import simbench_build
        return build_from_reference(cls, class_kwargs, getattr(simbench_build, prepare) if prepare else None)
    from . import native_definition, simbench_build
    if native_definition.has_recipe(ref_path) and not PREFER_RECORDED:
        # the native, parameterised builder (rule tables of opfgym_amd/native_definition.py): any constructor
        # arguments, on the synthetic stand-in grids.  (definition.PREFER_RECORDED = True: the recorded files instead.)
        return native_definition.build(ref_path, class_kwargs, grid_seed,
                                       getattr(simbench_build, prepare) if prepare else None)
    key = request_key(ref_path, class_kwargs, grid_seed, prepare)
    idx = _index()
    if key not in idx:
        raise ImportError(
            f'{ref_path}({class_kwargs}) is defined by the reference package `opfgym`, which is not importable here; '
            f'it has no native recipe (opfgym_amd/native_definition.py) and no recorded definition for these arguments '
            f'(opfgym_amd/definitions/index.json).  Install opfgym, or record one with tests/golden/make_definitions.py, '
            f'or build the environment from your own net and keys with BatchedOpfEnv(net, action_keys, observation_keys, ...).')
    return load(os.path.join(DEF_DIR, idx[key]))
