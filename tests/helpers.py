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
