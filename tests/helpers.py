"""Shared helpers for the test-suite (test code only)."""
import numpy as np

from opfgym_amd.case import net_to_case
from oracle import pf_oracle as po


def random_injections(net, case, B, seed, lo=0.2, hi=1.1):
    """B instances: every unit's P/Q scaled by its own uniform factor; returns
    p_inj, q_inj [B, nb] in p.u. (generation - demand)."""
    rng = np.random.default_rng(seed)
    base = case.base_mva
    nb = case.nb
    p = np.zeros((B, nb))
    q = np.zeros((B, nb))
    for tbl, sign in (('load', -1.0), ('sgen', 1.0), ('storage', -1.0), ('gen', 1.0)):
        df = net[tbl]
        if not len(df):
            continue
        bus = np.array([case.bus_lookup[int(b)] for b in df['bus']])
        f = rng.uniform(lo, hi, (B, len(df)))
        pv = df['p_mw'].to_numpy(float)[None, :] * f * sign / base
        np.add.at(p, (slice(None), bus), pv)
        if tbl != 'gen':
            f2 = rng.uniform(lo, hi, (B, len(df)))
            qv = df['q_mvar'].to_numpy(float)[None, :] * f2 * sign / base
            np.add.at(q, (slice(None), bus), qv)
    return p, q


def oracle_batch(case, p, q, **kw):
    """Solve every row with the SciPy oracle."""
    B = p.shape[0]
    vm = np.zeros((B, case.nb))
    va = np.zeros((B, case.nb))
    load = np.zeros((B, case.nbr))
    sref = np.zeros((B, int((case.bus_type == 3).sum()), 2))
    conv = np.zeros(B, bool)
    its = np.zeros(B, int)
    for b in range(B):
        outage = kw.get('outage')
        st = None
        if outage is not None and outage[b] >= 0:
            st = np.ones(case.nbr)
            st[outage[b]] = 0.0
        sol = po.solve_case(case, p[b], q[b], qg_min=kw.get('qg_min'), qg_max=kw.get('qg_max'),
                            qd_bus=-q[b], enforce_q_lims=kw.get('enforce_q_lims', False),
                            tol=kw.get('tol', 1e-8), max_it=kw.get('max_it', 10), br_status=st)
        v = sol['V']
        vm[b], va[b] = np.abs(v), np.angle(v)
        conv[b], its[b] = sol['converged'], sol['iterations']
        load[b] = po.branch_results(case, v, st)['loading_percent']
        s = v * np.conj(sol['ybus'] @ v)
        ref = np.flatnonzero(case.bus_type == 3)
        sref[b, :, 0] = s.real[ref] - p[b, ref]
        sref[b, :, 1] = s.imag[ref] - q[b, ref]
    return dict(vm=vm, va=va, loading=load, s_ref=sref, converged=conv, iterations=its)


def non_bridge_branches(case):
    """Branches whose outage leaves every bus connected to a REF bus."""
    out = []
    ref = np.flatnonzero(case.bus_type == 3)
    for k in range(case.nbr):
        adj = [[] for _ in range(case.nb)]
        for m in range(case.nbr):
            if m != k:
                adj[case.f[m]].append(case.t[m])
                adj[case.t[m]].append(case.f[m])
        seen = np.zeros(case.nb, bool)
        stack = list(ref)
        seen[ref] = True
        while stack:
            a = stack.pop()
            for b in adj[a]:
                if not seen[b]:
                    seen[b] = True
                    stack.append(b)
        if seen.all():
            out.append(k)
    return np.array(out, dtype=np.int32)


def ieee14_ppc():
    """IEEE 14-bus test system in pypower matrix form (public textbook data, MATPOWER `case14`;
    0-based bus numbers) and its published power-flow solution (MATPOWER `runpf(case14)`: voltage
    magnitudes to 3 decimals, angles in degrees, slack generation in MW / MVAr)."""
    z = 0.0
    bus = np.array([      # bus type Pd Qd Gs Bs area Vm Va baseKV zone Vmax Vmin
        [0, 3, 0.0, 0.0, z, z, 1, 1.060, 0.0, 0, 1, 1.06, 0.94],
        [1, 2, 21.7, 12.7, z, z, 1, 1.045, 0.0, 0, 1, 1.06, 0.94],
        [2, 2, 94.2, 19.0, z, z, 1, 1.010, 0.0, 0, 1, 1.06, 0.94],
        [3, 1, 47.8, -3.9, z, z, 1, 1.0, 0.0, 0, 1, 1.06, 0.94],
        [4, 1, 7.6, 1.6, z, z, 1, 1.0, 0.0, 0, 1, 1.06, 0.94],
        [5, 2, 11.2, 7.5, z, z, 1, 1.070, 0.0, 0, 1, 1.06, 0.94],
        [6, 1, 0.0, 0.0, z, z, 1, 1.0, 0.0, 0, 1, 1.06, 0.94],
        [7, 2, 0.0, 0.0, z, z, 1, 1.090, 0.0, 0, 1, 1.06, 0.94],
        [8, 1, 29.5, 16.6, z, 19.0, 1, 1.0, 0.0, 0, 1, 1.06, 0.94],
        [9, 1, 9.0, 5.8, z, z, 1, 1.0, 0.0, 0, 1, 1.06, 0.94],
        [10, 1, 3.5, 1.8, z, z, 1, 1.0, 0.0, 0, 1, 1.06, 0.94],
        [11, 1, 6.1, 1.6, z, z, 1, 1.0, 0.0, 0, 1, 1.06, 0.94],
        [12, 1, 13.5, 5.8, z, z, 1, 1.0, 0.0, 0, 1, 1.06, 0.94],
        [13, 1, 14.9, 5.0, z, z, 1, 1.0, 0.0, 0, 1, 1.06, 0.94]])
    br = [  # from to r x b tap   (1-based as published)
        (1, 2, 0.01938, 0.05917, 0.0528, 0), (1, 5, 0.05403, 0.22304, 0.0492, 0), (2, 3, 0.04699, 0.19797, 0.0438, 0),
        (2, 4, 0.05811, 0.17632, 0.0340, 0), (2, 5, 0.05695, 0.17388, 0.0346, 0), (3, 4, 0.06701, 0.17103, 0.0128, 0),
        (4, 5, 0.01335, 0.04211, 0.0, 0), (4, 7, 0.0, 0.20912, 0.0, 0.978), (4, 9, 0.0, 0.55618, 0.0, 0.969),
        (5, 6, 0.0, 0.25202, 0.0, 0.932), (6, 11, 0.09498, 0.19890, 0.0, 0), (6, 12, 0.12291, 0.25581, 0.0, 0),
        (6, 13, 0.06615, 0.13027, 0.0, 0), (7, 8, 0.0, 0.17615, 0.0, 0), (7, 9, 0.0, 0.11001, 0.0, 0),
        (9, 10, 0.03181, 0.08450, 0.0, 0), (9, 14, 0.12711, 0.27038, 0.0, 0), (10, 11, 0.08205, 0.19207, 0.0, 0),
        (12, 13, 0.22092, 0.19988, 0.0, 0), (13, 14, 0.17093, 0.34802, 0.0, 0)]
    branch = np.array([[f - 1, t - 1, r, x, b, 0, 0, 0, tap, 0, 1, -360, 360] for f, t, r, x, b, tap in br], dtype=float)
    gen = np.array([      # bus Pg Qg Qmax Qmin Vg mBase status
        [0, 232.4, -16.9, 10.0, 0.0, 1.060, 100, 1], [1, 40.0, 42.4, 50.0, -40.0, 1.045, 100, 1],
        [2, 0.0, 23.4, 40.0, 0.0, 1.010, 100, 1], [5, 0.0, 12.2, 24.0, -6.0, 1.070, 100, 1],
        [7, 0.0, 17.4, 24.0, -6.0, 1.090, 100, 1]])
    published = dict(
        vm=np.array([1.060, 1.045, 1.010, 1.018, 1.020, 1.070, 1.062, 1.090, 1.056, 1.051, 1.057, 1.055, 1.050, 1.036]),
        va_deg=np.array([0.0, -4.983, -12.725, -10.313, -8.774, -14.221, -13.360, -13.360, -14.939, -15.097,
                         -14.791, -15.076, -15.156, -16.034]),
        p_slack_mw=232.39, q_slack_mvar=-16.55, losses_mw=13.39)
    return 100.0, bus, branch, gen, published
