"""Shared helpers for the test-suite (test code only).

The two sides of every parity test are built independently: the product side from
`opfgym_amd.case.net_to_case` (branch admittance stamps, product bus order), the oracle side from
`oracle.pd2ppc.build_ppc` (r/x/b/tap/shift branch table, its own bus order, auxiliary buses).  The
only thing they share is the NET's bus / element indices, through which results are matched."""
import copy

import numpy as np

from oracle import pd2ppc
from oracle import pf_oracle as po


def random_injections(net, case, B, seed, lo=0.2, hi=1.1):
    """B instances: every unit's P/Q scaled by its own uniform factor; returns
    p_inj, q_inj [B, nb] in p.u. (generation - demand) in the product's bus order."""
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


class OracleSide:
    """The oracle's own case of `net` plus the index maps product order -> oracle order (via the net)."""

    def __init__(self, net, case):
        self.net, self.case = net, case
        self.ppc = pd2ppc.build_ppc(net)
        first = {}
        for b, i in case.bus_lookup.items():
            first.setdefault(i, b)
        tables = {0: 'line', 1: 'trafo', 2: 'trafo3w', 3: 'impedance', 4: 'switch', 5: 'xward'}
        sides = case.br_side if case.br_side is not None else np.zeros(case.nbr, int)
        self.br_map = np.array([self.ppc.branch_of(tables[int(kd)], int(e), ('hv', 'mv', 'lv')[int(sd)] if kd == 2 else '')
                                for kd, e, sd in zip(case.br_kind, case.br_elem, sides)])
        assert (self.br_map >= 0).all()
        # product bus -> ppc bus; the auxiliary buses (star points of three-winding transformers, internal buses of xwards)
        # have no net index: they are matched through the branch that ends at them (hv winding -> star, xward impedance -> source)
        aux = {}
        for k, (kd, sd) in enumerate(zip(case.br_kind, sides)):
            if (kd == 2 and sd == 0) or kd == 5:
                aux[int(case.t[k])] = int(self.ppc.t[self.br_map[k]])
        self.bus_map = np.array([self.ppc.bus_lookup[first[i]] if i in first else aux[i] for i in range(case.nb)])
        self.ref = np.flatnonzero(case.bus_type == 3)

    def solve(self, p, q, outage=-1, qg_min=None, qg_max=None, enforce_q_lims=False, tol=1e-8, max_it=10, init='flat'):
        """One instance: bus injections p, q [nb] (p.u., product order; generator active power included
        in p) -> results in product order."""
        ppc, case = copy.copy(self.ppc), self.case
        base = ppc.base_mva
        ppc.pd, ppc.qd = np.zeros(ppc.nb), np.zeros(ppc.nb)
        ppc.pd[self.bus_map] = -p * base
        ppc.qd[self.bus_map] = -q * base
        ppc.g_p = np.zeros(len(ppc.g_bus))
        if qg_min is not None:                 # per product bus, shared equally by the generators of the bus
            ppc.g_qmin, ppc.g_qmax = ppc.g_qmin.copy(), ppc.g_qmax.copy()
            inv = {int(b): i for i, b in enumerate(self.bus_map)}
            n_at = np.bincount(ppc.g_bus, minlength=ppc.nb)
            for g in range(len(ppc.g_bus)):
                # (every generator row that regulates a bus: the net's generators, the two ends of DC lines, the internal sources of xwards)
                if ppc.g_table[g] != 'ext_grid' and int(ppc.g_bus[g]) in inv:
                    i = inv[int(ppc.g_bus[g])]
                    ppc.g_qmin[g] = max(qg_min[i] * base / n_at[ppc.g_bus[g]], -1e9)
                    ppc.g_qmax[g] = min(qg_max[i] * base / n_at[ppc.g_bus[g]], 1e9)
        status = None
        if outage is not None and outage >= 0:
            status = ppc.status.copy()
            status[self.br_map[outage]] = 0
        sol = po.solve(ppc, enforce_q_lims=enforce_q_lims, tol=tol, max_it=max_it, status=status, init=init)
        v = sol['V'][self.bus_map]
        ld = po.loading_percent(ppc, self.net, sol['V'], sol['status'])
        # (trafo3w: the transformer's value on each of its three windings; impedances and switch branches: no rating, 0 %)
        loading = np.array([ld[('line', 'trafo', 'trafo3w')[int(kd)]][int(e)] if kd < 3 else 0.0 for kd, e in zip(case.br_kind, case.br_elem)])
        s_calc = (sol['V'] * np.conj(sol['ybus'] @ np.nan_to_num(sol['V'])))[self.bus_map]
        sref = np.stack([s_calc.real[self.ref] - p[self.ref], s_calc.imag[self.ref] - q[self.ref]], axis=1)
        return dict(vm=np.abs(v), va=np.angle(v), loading=loading, s_ref=sref, converged=sol['converged'],
                    iterations=sol['iterations'], V=v)


def oracle_batch(net, case, p, q, **kw):
    """Solve every row with the oracle, built from `net` by the oracle's own converter."""
    side = OracleSide(net, case)
    B = p.shape[0]
    out = dict(vm=np.zeros((B, case.nb)), va=np.zeros((B, case.nb)), loading=np.zeros((B, case.nbr)),
               s_ref=np.zeros((B, len(side.ref), 2)), converged=np.zeros(B, bool), iterations=np.zeros(B, int))
    outage = kw.pop('outage', None)
    for b in range(B):
        r = side.solve(p[b], q[b], outage=-1 if outage is None else int(outage[b]), **kw)
        for k in out:
            out[k][b] = r[k]
    return out


def oracle_ppc_solve(base, bus, branch, gen, **kw):
    """A published case given as pypower matrices, solved by the oracle (its own matrix reader)."""
    ppc = pd2ppc.ppc_from_matrices(base, bus, branch, gen)
    sol = po.solve(ppc, **kw)
    sol['ppc'] = ppc
    return sol


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


def ieee30_ppc():
    """IEEE 30-bus test system (PSTCA common-data-format case: 41 branches, four off-nominal taps, two bus
    shunts, five PV buses; 0-based bus numbers) in pypower matrix form.  `published`: values of its solved load
    flow as MATPOWER prints them for `case_ieee30` (no reactive limits enforced), quoted to the digits asserted —
    only quantities known independently of this repository are listed, a subset of buses; a single wrong
    parameter among the 41 branches would move every one of them."""
    z = 0.0
    loads = {2: (21.7, 12.7), 3: (2.4, 1.2), 4: (7.6, 1.6), 5: (94.2, 19.0), 7: (22.8, 10.9), 8: (30.0, 30.0), 10: (5.8, 2.0),
             12: (11.2, 7.5), 14: (6.2, 1.6), 15: (8.2, 2.5), 16: (3.5, 1.8), 17: (9.0, 5.8), 18: (3.2, 0.9), 19: (9.5, 3.4),
             20: (2.2, 0.7), 21: (17.5, 11.2), 23: (3.2, 1.6), 24: (8.7, 6.7), 26: (3.5, 2.3), 29: (2.4, 0.9), 30: (10.6, 1.9)}
    gens = {1: (0.0, 1.060, 3), 2: (40.0, 1.043, 2), 5: (0.0, 1.010, 2), 8: (0.0, 1.010, 2), 11: (0.0, 1.082, 2), 13: (0.0, 1.071, 2)}
    shunt = {10: 19.0, 24: 4.3}
    bus = np.array([[i - 1, gens[i][2] if i in gens else 1, *loads.get(i, (0.0, 0.0)), z, shunt.get(i, 0.0), 1,
                     gens[i][1] if i in gens else 1.0, 0.0, 132, 1, 1.1, 0.9] for i in range(1, 31)])
    br = [  # from to r x b tap   (1-based as published)
        (1, 2, 0.0192, 0.0575, 0.0528, 0), (1, 3, 0.0452, 0.1652, 0.0408, 0), (2, 4, 0.0570, 0.1737, 0.0368, 0),
        (3, 4, 0.0132, 0.0379, 0.0084, 0), (2, 5, 0.0472, 0.1983, 0.0418, 0), (2, 6, 0.0581, 0.1763, 0.0374, 0),
        (4, 6, 0.0119, 0.0414, 0.0090, 0), (5, 7, 0.0460, 0.1160, 0.0204, 0), (6, 7, 0.0267, 0.0820, 0.0170, 0),
        (6, 8, 0.0120, 0.0420, 0.0090, 0), (6, 9, 0.0, 0.2080, 0.0, 0.978), (6, 10, 0.0, 0.5560, 0.0, 0.969),
        (9, 11, 0.0, 0.2080, 0.0, 0), (9, 10, 0.0, 0.1100, 0.0, 0), (4, 12, 0.0, 0.2560, 0.0, 0.932),
        (12, 13, 0.0, 0.1400, 0.0, 0), (12, 14, 0.1231, 0.2559, 0.0, 0), (12, 15, 0.0662, 0.1304, 0.0, 0),
        (12, 16, 0.0945, 0.1987, 0.0, 0), (14, 15, 0.2210, 0.1997, 0.0, 0), (16, 17, 0.0524, 0.1923, 0.0, 0),
        (15, 18, 0.1073, 0.2185, 0.0, 0), (18, 19, 0.0639, 0.1292, 0.0, 0), (19, 20, 0.0340, 0.0680, 0.0, 0),
        (10, 20, 0.0936, 0.2090, 0.0, 0), (10, 17, 0.0324, 0.0845, 0.0, 0), (10, 21, 0.0348, 0.0749, 0.0, 0),
        (10, 22, 0.0727, 0.1499, 0.0, 0), (21, 22, 0.0116, 0.0236, 0.0, 0), (15, 23, 0.1000, 0.2020, 0.0, 0),
        (22, 24, 0.1150, 0.1790, 0.0, 0), (23, 24, 0.1320, 0.2700, 0.0, 0), (24, 25, 0.1885, 0.3292, 0.0, 0),
        (25, 26, 0.2544, 0.3800, 0.0, 0), (25, 27, 0.1093, 0.2087, 0.0, 0), (28, 27, 0.0, 0.3960, 0.0, 0.968),
        (27, 29, 0.2198, 0.4153, 0.0, 0), (27, 30, 0.3202, 0.6027, 0.0, 0), (29, 30, 0.2399, 0.4533, 0.0, 0),
        (8, 28, 0.0636, 0.2000, 0.0428, 0), (6, 28, 0.0169, 0.0599, 0.0130, 0)]
    branch = np.array([[f - 1, t - 1, r, x, b, 0, 0, 0, tap, 0, 1, -360, 360] for f, t, r, x, b, tap in br], dtype=float)
    gen = np.array([[b - 1, pg, 0.0, 1e4, -1e4, vg, 100, 1] for b, (pg, vg, _) in gens.items()], dtype=float)
    published = dict(
        vm={2: 1.021, 3: 1.012, 6: 1.002, 29: 0.992}, vm_tol=6e-4,                 # buses 3, 4, 7, 30 (0-based keys)
        va_deg={2: -7.53, 3: -9.28, 6: -12.86}, va_tol=7e-3,
        p_slack_mw=260.95, losses_mw=17.55, s_tol=0.011,
        qg_mvar={3: 37.22, 4: 16.18, 5: 10.63})                                    # generators at buses 8, 11, 13 (rows of `gen`)
    return 100.0, bus, branch, gen, published


def dense_ppc(n, seed=0):
    """A complete graph of `n` buses in pypower matrix form (every pair of buses joined by a line): the elimination
    has no independent vertices at all, every level holds one pivot — the whole matrix is the "dense tail" of the
    wave-team kernels (tail sizes up to and beyond their 32-pivot register chain)."""
    rng = np.random.default_rng(seed)
    z = 0.0
    bus = np.array([[i, 3 if i == 0 else (2 if i in (3, 7) else 1), rng.uniform(1, 6), rng.uniform(0.2, 2), z, z, 1,
                     1.02 if i in (0, 3, 7) else 1.0, 0.0, 110, 1, 1.1, 0.9] for i in range(n)])
    pairs = [(f, t) for f in range(n) for t in range(f + 1, n)]
    branch = np.array([[f, t, rng.uniform(0.02, 0.08), rng.uniform(0.1, 0.4), rng.uniform(0, 0.02), 0, 0, 0, 0, 0, 1, -360, 360]
                       for f, t in pairs])
    gen = np.array([[0, 0.0, 0.0, 1e4, -1e4, 1.02, 100, 1], [3, 12.0, 0.0, 1e4, -1e4, 1.02, 100, 1], [7, 9.0, 0.0, 1e4, -1e4, 1.02, 100, 1]])
    return 100.0, bus, branch, gen


def _ppc(base_kv, bus_rows, br_rows, gen_rows):
    """pypower matrices from compact rows: bus (type, Pd, Qd, Vm), branch (from, to, r, x, b; 1-based),
    gen (bus (1-based), Pg, Vg, Qmax, Qmin)."""
    bus = np.array([[i, t, pd_, qd_, 0.0, 0.0, 1, vm, 0.0, base_kv, 1, 1.1, 0.9] for i, (t, pd_, qd_, vm) in enumerate(bus_rows)])
    branch = np.array([[f - 1, t - 1, r, x, b, 0, 0, 0, 0, 0, 1, -360, 360] for f, t, r, x, b in br_rows], dtype=float)
    gen = np.array([[b - 1, pg, 0.0, qmax, qmin, vg, 100, 1] for b, pg, vg, qmax, qmin in gen_rows], dtype=float)
    return 100.0, bus, branch, gen


def published_cases():
    """Textbook systems with published load-flow solutions (public data, typed in by hand; 0-based bus
    numbers).  name -> (base, bus, branch, gen, published): `published` holds the printed numbers and the
    tolerance their printed precision allows.
      * gs4:  Grainger & Stevenson, "Power System Analysis", 4-bus example (MATPOWER `case4gs`);
      * ww6:  Wood & Wollenberg, "Power Generation, Operation and Control", 6-bus system (`case6ww`);
      * sea5: Stagg & El-Abiad, "Computer Methods in Power System Analysis", 5-bus system (its generator at
              bus 2 is a fixed P,Q source: booked as negative demand).
    (IEEE 14-bus: `ieee14_ppc`; WSCC 9-bus: opfgym_amd.grids.case9.)"""
    gs4 = _ppc(230.0, [(3, 50, 30.99, 1.0), (1, 170, 105.35, 1.0), (1, 200, 123.94, 1.0), (2, 80, 49.58, 1.02)],
               [(1, 2, 0.01008, 0.0504, 0.1025), (1, 3, 0.00744, 0.0372, 0.0775), (2, 4, 0.00744, 0.0372, 0.0775),
                (3, 4, 0.01272, 0.0636, 0.1275)], [(1, 0, 1.0, 1e4, -1e4), (4, 318, 1.02, 1e4, -1e4)])
    ww6 = _ppc(230.0, [(3, 0, 0, 1.05), (2, 0, 0, 1.05), (2, 0, 0, 1.07), (1, 70, 70, 1.0), (1, 70, 70, 1.0), (1, 70, 70, 1.0)],
               [(1, 2, 0.1, 0.2, 0.04), (1, 4, 0.05, 0.2, 0.04), (1, 5, 0.08, 0.3, 0.06), (2, 3, 0.05, 0.25, 0.06),
                (2, 4, 0.05, 0.1, 0.02), (2, 5, 0.1, 0.3, 0.04), (2, 6, 0.07, 0.2, 0.05), (3, 5, 0.12, 0.26, 0.05),
                (3, 6, 0.02, 0.1, 0.02), (4, 5, 0.2, 0.4, 0.08), (5, 6, 0.1, 0.3, 0.06)],
               [(1, 0, 1.05, 1e4, -1e4), (2, 50, 1.05, 1e4, -1e4), (3, 60, 1.07, 1e4, -1e4)])
    sea5 = _ppc(100.0, [(3, 0, 0, 1.06), (1, 20 - 40, 10 - 30, 1.0), (1, 45, 15, 1.0), (1, 40, 5, 1.0), (1, 60, 10, 1.0)],
                [(1, 2, 0.02, 0.06, 0.06), (1, 3, 0.08, 0.24, 0.05), (2, 3, 0.06, 0.18, 0.04), (2, 4, 0.06, 0.18, 0.04),
                 (2, 5, 0.04, 0.12, 0.03), (3, 4, 0.01, 0.03, 0.02), (4, 5, 0.08, 0.24, 0.05)], [(1, 0, 1.06, 1e4, -1e4)])
    return {
        'gs4': gs4 + (dict(vm=[1.0, 0.982, 0.969, 1.02], vm_tol=6e-4, va_deg=[0.0, -0.976, -1.872, 1.523], va_tol=6e-4,
                           pg={0: 186.81}, qg={0: 114.50, 1: 181.43}, s_tol=6e-3),),
        'ww6': ww6 + (dict(vm=[1.05, 1.05, 1.07, 0.9894, 0.9854, 1.0044], vm_tol=6e-5,
                           va_deg=[0.0, -3.67, -4.27, -4.20, -5.28, -5.95], va_tol=6e-3,
                           pg={0: 107.87}, qg={0: 15.96, 1: 74.36, 2: 89.63}, s_tol=1.1e-2),),
        'sea5': sea5 + (dict(vm=[1.06, 1.0474, 1.0242, 1.0236, 1.0179], vm_tol=1.1e-4,
                             va_deg=[0.0, -2.806, -4.997, -5.329, -6.150], va_tol=1.1e-3,
                             pg={0: 129.59}, qg={0: -7.42}, s_tol=1.1e-2),),
    }


def resonant_leaf_ppc(n_chain=4):
    """A case on which STATIC pivoting breaks down although the Jacobian is regular: a chain of `n_chain` buses behind the
    slack and, at its end, a leaf bus without load whose shunt capacitor compensates HALF of its line's susceptance
    (line x = 0.1, r = 0: B_ij = 10; shunt +5 p.u.).  At the flat start the leaf's own 2x2 diagonal Jacobian block
    [[dP/dth, dP/dlnV], [dQ/dth, dQ/dlnV]] = [[10, 0], [0, 0]] is exactly singular — the reactive balance of that bus does
    not depend on ITS voltage to first order —, while the full matrix (the coupling block to the neighbour is -10 I) is
    not: SuperLU pivots on the coupling entries; a minimum-degree block elimination takes the leaf first and divides by
    zero.  pypower matrices (0-based): (base, bus, branch, gen), the leaf is the last bus."""
    nb = n_chain + 2
    z = 0.0
    bus = np.array([[i, 3 if i == 0 else 1, 0.0 if i in (0, nb - 1) else 20.0, 0.0 if i in (0, nb - 1) else 5.0, z, z, 1,
                     1.0, 0.0, 110, 1, 1.1, 0.9] for i in range(nb)])
    bus[nb - 1, 5] = 5.0 * 100.0                               # Bs [MVAr at 1 p.u.]: +5 p.u. on base 100
    branch = np.array([[i, i + 1, 0.01 if i < nb - 2 else 0.0, 0.1, 0.0, 0, 0, 0, 0, 0, 1, -360, 360] for i in range(nb - 1)], dtype=float)
    gen = np.array([[0, 0.0, 0.0, 1e4, -1e4, 1.0, 100, 1]])
    return 100.0, bus, branch, gen


# Developer switches of the library (include/opfx_debug.h) for the GPU tests.  The package never reads the process
# environment for them; this harness does, in ONE place: a test run may be steered as a whole from outside
# (OPFX_KERNEL_V1=1 python -m pytest ... re-runs the solver tests on the first-generation kernel,
# tests/test_gpu_solve.py::test_first_generation_kernel_still_correct), individual tests add their own members.
DEBUG_OVERRIDES = {}


def harness_debug(**members):
    from opfgym_amd import capi
    d = capi.debug_from_env()
    for k, v in {**DEBUG_OVERRIDES, **members}.items():
        setattr(d, k, int(v))
    return d
