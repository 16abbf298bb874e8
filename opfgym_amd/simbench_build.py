"""Host-side helpers around the SimBench time series: the train / validation / test split
(`data_split.py:5-59`) and the time observation (`time_observation.py:4-22`) that the environment needs at
run time, plus two helpers for the synthetic stand-in grids.  (Grid preparation itself — scaling columns,
system constraints, profile repair, limits from the profiles, `build_simbench_net.py:5-97` — is the
reference's and reaches this package as data, see opfgym_amd/definition.py.)
"""
from __future__ import annotations

import numpy as np



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
    # limit / mean / std columns follow the profiles: recomputed by the reference's own function where the
    # reference is at hand (recording a definition), by the native grid preparation otherwise
    try:
        from opfgym.simbench.build_simbench_net import set_constraints_from_profiles
    except Exception:
        from .native_definition import ranges_from_profiles as set_constraints_from_profiles
    if 'scaling' in net.sgen.columns and 'scaling' in net.load.columns and 'scaling' in net.storage.columns:
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


def add_switched_shunts(net, profiles):
    """Stand-in helper (no counterpart in the reference's grids): three shunts in steps — a capacitor bank, a bank with
    losses and a reactor — at the 4th, 8th and 12th bus, with the limit columns an `('shunt', 'step')` action key needs
    (`min_step` / `max_step` and their `min_min_` / `max_max_` twins, opf_env.py:439-446).  In place."""
    import pandas as pd
    buses = net.bus.index[[3, 7, 11]]
    sh = pd.DataFrame(dict(bus=np.asarray(buses, dtype=np.int64), p_mw=[0.0, 0.5, 0.0], q_mvar=[-8.0, -5.0, 6.0],
                           vn_kv=net.bus.vn_kv.loc[buses].to_numpy(float), step=[1, 0, 2], max_step=[4, 3, 2],
                           min_step=[0, 0, 0], in_service=True))
    sh['max_max_step'] = sh['max_step']
    sh['min_min_step'] = sh['min_step']
    net['shunt'] = sh
    return net, profiles


def split_busbars(net, profiles, buses=(7, 11)):
    """Stand-in helper (the synthetic grids have single busbars): the given substation buses become two busbars with a
    COUPLER — a closed bus-bus switch — between them; every second line end moves to the new bar.  Coupler closed: the grid
    as it was; open: the two bars hang together through the rest of the meshed grid only.  The couplers are the last rows of
    the switch table.  In place."""
    from . import net as ppn
    for b in buses:
        ends = [(i, 'from_bus') for i in net.line.index[net.line.from_bus == b]] + \
               [(i, 'to_bus') for i in net.line.index[net.line.to_bus == b]]
        assert len(ends) >= 4, f'bus {b}: a busbar split needs at least four line ends'
        new = ppn.create_bus(net, vn_kv=float(net.bus.vn_kv.at[b]))
        for c in net.bus.columns:
            if c != 'name':
                net.bus.at[new, c] = net.bus.at[b, c]
        for i, side in ends[1::2]:
            net.line.at[i, side] = new
        ppn.create_switch(net, int(b), int(new), 'b', closed=True)
    return net, profiles


def share_generator_buses(net, profiles):
    """Stand-in helper (the synthetic grids carry one generator per bus): more generators on buses that already have one —
    a second on the bus of generator 0, a second and a third on the bus of generator 1, one OUT OF SERVICE on the bus of
    generator 2, and one on the bus of the first ext_grid (a REF bus).  pypower's `pfsoln` splits the reactive power
    generated at such a bus among its generators by their reactive ranges (`case.generator_dispatch`); `res_gen.q_mvar`
    and the cost rows of these generators (objective.py:48-54) depend on that split.  Time series of the new rows: an
    existing generator's, scaled.  In place."""
    from . import net as ppn
    gen = net.gen
    g_bus = [int(b) for b in gen.bus.iloc[:3]]
    new = [(g_bus[0], 8.0, 0, True), (g_bus[1], 12.0, 1, True), (g_bus[1], 6.0, 2, True), (g_bus[2], 9.0, 0, False),
           (int(net.ext_grid.bus.iloc[0]), 5.0, 1, True)]
    key = ('gen', 'p_mw')
    df = profiles[key] if profiles is not None and key in profiles else None
    scaling = float(gen.scaling.iloc[0]) if 'scaling' in gen.columns else 1.0
    added = []
    for bus, p_mw, like, on in new:
        vm = float(gen.vm_pu[gen.bus == bus].iloc[0]) if (gen.bus == bus).any() else float(net.ext_grid.vm_pu.iloc[0])
        idx = ppn.create_gen(net, bus, p_mw, vm_pu=vm, in_service=on, scaling=scaling)
        added.append(idx)
        if df is not None:
            src = df.columns[like]
            df[idx] = df[src].to_numpy() / max(float(df[src].max()), 1e-9) * p_mw
    ppn.finalize(net)
    if df is not None and 'max_max_p_mw' in net.gen.columns:
        # applied to a grid `build_simbench_net` has already worked on: the columns it derives from the time series
        # (build_simbench_net.py:68-81) for the new rows, the same way
        for idx in added:
            net.gen.at[idx, 'max_max_p_mw'] = float(df[idx].max()) * scaling
            net.gen.at[idx, 'min_min_p_mw'] = float(df[idx].min()) * scaling
            net.gen.at[idx, 'mean_p_mw'] = float(df[idx].mean())
            net.gen.at[idx, 'std_dev_p_mw'] = float(df[idx].std())
    if df is not None and hasattr(profiles, 'rel'):
        profiles.rel.pop(key, None)               # (the factored form no longer covers the table)
    return net, profiles


def shared_bus_reactive_setup(net):
    """Reactive ranges and reactive cost terms for the generators of `share_generator_buses` (applied AFTER the problem
    definition, which for EcoDispatch zeroes every reactive range, eco_dispatch.py:84-88): different ranges on a shared
    bus, a zero range beside non-zero ones, a bus whose summed range binds under `enforce_q_lims`, and `cq1` / `cq2`
    prices on the generators' polynomial cost rows so that the objective reads `res_gen.q_mvar` per generator.  In place."""
    lo = [-10.0, -20.0, -15.0, -1.0, -4.0, -6.0, 0.0, -5.0, -5.0]
    hi = [4.0, 25.0, 20.0, 0.15, 1.5, 6.0, 0.0, 5.0, 8.0]
    n = len(net.gen)
    assert n == len(lo), 'shared_bus_reactive_setup expects the generators of hv-small + share_generator_buses'
    net.gen['min_q_mvar'] = lo
    net.gen['max_q_mvar'] = hi
    pc = net.poly_cost
    if not len(pc):
        return net
    rows = pc.index[pc.et == 'gen']
    pos = net.gen.index.get_indexer([int(e) for e in pc.loc[rows, 'element']])
    pc.loc[rows, 'cq1_eur_per_mvar'] = 0.02 * (1.0 + pos)
    pc.loc[rows, 'cq2_eur_per_mvar2'] = 0.001 * (1.0 + (pos % 3))
    return net
