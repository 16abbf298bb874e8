"""Grid + profile preparation for the benchmark environments.

Host-side, once per environment.  Same behaviour as the reference's
`opfgym/simbench/build_simbench_net.py:5-97`, `data_split.py:5-59` and
`time_observation.py:4-22`, working on :class:`opfgym_amd.net.Net` tables (or a
real pandapowerNet) and a profile dict.  The grid itself comes from
`opfgym_amd.grids.get_grid` (synthetic stand-ins: SimBench is not available
here) or from the caller.
"""
from __future__ import annotations

import numpy as np

from . import grids


def build_simbench_net(simbench_network_name, gen_scaling=1.0, load_scaling=1.0,
                       storage_scaling=1.0, voltage_band=0.05, max_loading=80,
                       grid_seed=0, net=None, profiles=None, *args, **kwargs):
    """build_simbench_net.py:5-23: scaling columns, system constraints,
    profile repair, min/max/mean/std columns from the profiles."""
    if net is None:
        net, profiles = grids.get_grid(simbench_network_name, grid_seed)
    set_unit_scaling(net, gen_scaling, load_scaling, storage_scaling)
    set_system_constraints(net, voltage_band, max_loading)
    repair_simbench_profiles(net, profiles)
    set_constraints_from_profiles(net, profiles)
    return net, profiles


def set_unit_scaling(net, gen_scaling=1.0, load_scaling=1.0, storage_scaling=1.0):
    # build_simbench_net.py:26-31
    net.sgen['scaling'] = gen_scaling
    net.gen['scaling'] = gen_scaling
    net.load['scaling'] = load_scaling
    net.storage['scaling'] = storage_scaling


def set_system_constraints(net, voltage_band=None, max_loading=None):
    # build_simbench_net.py:34-42
    if voltage_band:
        net.bus['max_vm_pu'] = 1 + voltage_band
        net.bus['min_vm_pu'] = 1 - voltage_band
    if max_loading:
        net.line['max_loading_percent'] = max_loading
        net.trafo['max_loading_percent'] = max_loading


def repair_simbench_profiles(net, profiles):
    # build_simbench_net.py:45-64: negative sgen power -> 0; drop constant units
    sg = profiles[('sgen', 'p_mw')]
    sg[sg < 0.0] = 0.0
    for key in list(profiles.keys()):
        df = profiles[key]
        tbl = net[key[0]]
        is_equal = df.max(axis=0) == df.min(axis=0)
        tbl.drop(tbl[is_equal].index, inplace=True)
        df.drop(columns=df.columns[is_equal], inplace=True)


def set_constraints_from_profiles(net, profiles):
    # build_simbench_net.py:67-97
    for (unit_type, column), df in profiles.items():
        tbl = net[unit_type]
        if unit_type == 'storage':
            max_power = np.maximum(df.max(axis=0).abs(), df.min(axis=0).abs())
            tbl[f'max_max_{column}'] = max_power * tbl.scaling
            tbl[f'min_min_{column}'] = -max_power * tbl.scaling
        else:
            tbl[f'max_max_{column}'] = df.max(axis=0) * tbl.scaling
            tbl[f'min_min_{column}'] = df.min(axis=0) * tbl.scaling
        tbl[f'mean_{column}'] = df.mean(axis=0)
        tbl[f'std_dev_{column}'] = df.std(axis=0)
    diff = profiles[('load', 'p_mw')].sum(axis=1) - profiles[('sgen', 'p_mw')].sum(axis=1)
    net.ext_grid['max_max_p_mw'] = diff.max()
    net.ext_grid['min_min_p_mw'] = diff.min()
    net.ext_grid['mean_p_mw'] = diff.mean()
    load_q = profiles[('load', 'q_mvar')].sum(axis=1)
    net.ext_grid['max_max_q_mvar'] = load_q.max()
    net.ext_grid['min_min_q_mvar'] = load_q.min()
    net.ext_grid['mean_q_mvar'] = load_q.mean()


def gens_to_fixed_sgens(net, profiles):
    """Stand-in helper (no counterpart in the reference): VoltageControl asserts a grid without `gen`
    units (voltage_control.py:102), the HV stand-in grids carry PV generators — they become sgens with
    q = 0 that follow the generators' active-power profiles.  In place; a no-op without generators."""
    import pandas as pd
    if not len(net.gen):
        return net, profiles
    start = (int(net.sgen.index.max()) + 1) if len(net.sgen) else 0
    for k, (idx, row) in enumerate(net.gen.iterrows()):
        net.sgen.loc[start + k] = {c: row[c] if c in row else np.nan for c in net.sgen.columns}
        net.sgen.loc[start + k, 'q_mvar'] = 0.0
    if ('gen', 'p_mw') in profiles:
        df = profiles.pop(('gen', 'p_mw'))
        df.columns = [start + k for k in range(df.shape[1])]
        profiles[('sgen', 'p_mw')] = pd.concat([profiles[('sgen', 'p_mw')], df], axis=1)
        if hasattr(profiles, 'rel'):
            profiles.rel.pop(('sgen', 'p_mw'), None)
    net.gen = net.gen.iloc[0:0]
    net.sgen['bus'] = net.sgen['bus'].astype(np.int64)
    net.sgen['in_service'] = net.sgen['in_service'].astype(bool)
    set_constraints_from_profiles(net, profiles)
    return net, profiles


def non_islanding_lines(net):
    """Index values of the in-service lines whose outage leaves every energised bus connected to a
    slack (the contingency list of BASELINE config 5)."""
    from .case import KIND_LINE, net_to_case
    case = net_to_case(net)
    nb = case.nb
    ref = np.flatnonzero(case.bus_type == 3)
    coupled = (case.yft != 0) | (case.ytf != 0)
    out = []
    for k in np.flatnonzero((case.br_kind == KIND_LINE) & coupled):
        adj = [[] for _ in range(nb)]
        for m in np.flatnonzero(coupled):
            if m != k:
                adj[case.f[m]].append(case.t[m])
                adj[case.t[m]].append(case.f[m])
        seen = np.zeros(nb, bool)
        seen[ref] = True
        stack = list(ref)
        while stack:
            a = stack.pop()
            for b in adj[a]:
                if not seen[b]:
                    seen[b] = True
                    stack.append(b)
        if seen.all():
            out.append(int(net.line.index[case.br_elem[k]]))
    return np.array(out, dtype=np.int64)


def define_test_train_split(test_share=0.2, random_test_steps=False, validation_share=0.2,
                            random_validation_steps=False, **kwargs):
    """data_split.py:5-59: deterministic weekly blocks out of 35 136 steps
    (defaults: 6 720 test, 6 720 validation, 21 696 train; validation[0]=672)."""
    assert test_share + validation_share <= 1.0
    if random_test_steps:
        assert random_validation_steps
    n_points = 24 * 4 * 366
    all_steps = np.arange(n_points)
    one_week = 7 * 24 * 4
    test_weeks = np.array([], dtype=int)
    if test_share == 1.0:
        return all_steps, np.array([]), np.array([])
    if test_share == 0.0:
        test_steps = np.array([], dtype=int)
    elif random_test_steps:
        test_steps = np.random.choice(all_steps, int(n_points * test_share))
    else:
        test_weeks = np.linspace(0, 51, num=int(52 * test_share), dtype=int)
        test_steps = np.concatenate([np.arange(w * one_week, (w + 1) * one_week) for w in test_weeks])
    remaining = np.array(sorted(set(all_steps.tolist()) - set(test_steps.tolist())))
    if validation_share == 1.0:
        return np.array([]), all_steps, np.array([])
    if validation_share == 0.0:
        val_steps = np.array([], dtype=int)
    elif random_validation_steps:
        val_steps = np.random.choice(remaining, int(n_points * validation_share))
    else:
        if random_test_steps:
            test_weeks = np.array([], dtype=int)
        free_weeks = np.array(sorted(set(range(52)) - set(test_weeks.tolist())))
        pick = np.linspace(0, len(free_weeks) - 1, num=int(52 * validation_share), dtype=int)
        val_steps = np.concatenate([np.arange(w * one_week, (w + 1) * one_week)
                                    for w in free_weeks[pick]])
    train_steps = np.array(sorted(set(remaining.tolist()) - set(val_steps.tolist())))
    return test_steps, val_steps, train_steps


def get_simbench_time_observation(current_step, total_n_steps=24 * 4 * 366):
    """time_observation.py:4-22: sin/cos of day, week, year; vectorised over
    `current_step` (returns [..., 6])."""
    step = np.asarray(current_step)
    out = []
    for frame in (24 * 4, 7 * 24 * 4, total_n_steps):
        ang = 2 * np.pi * (step % frame) / frame
        out.append(np.sin(ang))
        out.append(np.cos(ang))
    return np.stack(out, axis=-1)
