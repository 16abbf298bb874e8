"""CPU oracle for the solver half of the hot path (SURVEY.md §8a rows P3–P7).

TEST INFRASTRUCTURE ONLY.  Nothing in the product package `opfgym_amd/` may
import or call this module; only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` do, and only as the checker / the timed CPU
baseline.

What it restates: the Newton-Raphson AC power flow that the reference reaches
through `pandapower.runpp(net, enforce_q_lims=True)` at
`/root/reference/opfgym/opf_env.py:703`.  pandapower (`>=2.13.1,<3.0`,
pyproject.toml:32) is a third-party dependency that is NOT vendored in
/root/reference and NOT installed here, so this file restates the published
pypower/MATPOWER algorithm it derives from: `makeYbus`, `makeSbus`, `newtonpf`
(polar full Newton on [ΔP(pv∪pq); ΔQ(pq)], ∞-norm stop at `tol`, ≤10
iterations, sparse direct solve — here SciPy SuperLU with partial pivoting, the
same class of solver pandapower uses without lightsim2grid), the
`enforce_q_lims` outer loop, `pfsoln` branch flows and pandapower's
`loading_percent` definitions.

PARITY UNPINNED against pandapower: the reference's own tests hold no
numerical power-flow result at all, and pandapower cannot be run
here.  This oracle is pinned instead by (tests/test_oracle_pf.py):
  * the closed-form two-bus solution,
  * the published WSCC 9-bus (`case9`) voltage profile,
  * the published IEEE 14-bus solution (off-nominal taps, bus shunt, four PV buses; |V| to the three
    published decimals, angles to 0.001 degree, slack generation and losses to 0.01 MW),
  * algebraic self-checks (mismatch < tol, power balance = losses).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
from scipy.sparse.linalg import splu

PQ, PV, REF = 1, 2, 3


class LoadflowNotConverged(Exception):
    """Stand-in for pandapower.powerflow.LoadflowNotConverged (opf_env.py:660)."""


# --------------------------------------------------------------------------
# P3: makeYbus / makeSbus
# --------------------------------------------------------------------------
def make_ybus(case, br_status=None) -> sp.csr_matrix:
    nb = case.nb
    s = np.ones(case.nbr) if br_status is None else np.asarray(br_status, float)
    rows = np.concatenate([case.f, case.f, case.t, case.t, np.arange(nb)])
    cols = np.concatenate([case.f, case.t, case.f, case.t, np.arange(nb)])
    vals = np.concatenate([case.yff * s, case.yft * s, case.ytf * s, case.ytt * s,
                           case.gs + 1j * case.bs])
    return sp.csr_matrix((vals, (rows, cols)), shape=(nb, nb))


# --------------------------------------------------------------------------
# P4: newtonpf
# --------------------------------------------------------------------------
def _ds_dv(ybus, v):
    ibus = ybus @ v
    diag_v = sp.diags(v)
    diag_i = sp.diags(ibus)
    diag_vn = sp.diags(v / np.abs(v))
    ds_dvm = diag_v @ np.conj(ybus @ diag_vn) + np.conj(diag_i) @ diag_vn
    ds_dva = 1j * diag_v @ np.conj(diag_i - ybus @ diag_v)
    return ds_dvm.tocsr(), ds_dva.tocsr()


def newtonpf(ybus, sbus, v0, pv, pq, tol=1e-8, max_it=10):
    """Restatement of pypower `newtonpf`.  Returns (V, converged, iterations,
    final ∞-norm of the mismatch)."""
    v = v0.astype(complex).copy()
    va, vm = np.angle(v), np.abs(v)
    pvpq = np.r_[pv, pq]
    npv, npq = len(pv), len(pq)

    def mismatch(v):
        mis = v * np.conj(ybus @ v) - sbus
        return np.r_[mis[pvpq].real, mis[pq].imag]

    f = mismatch(v)
    norm_f = np.max(np.abs(f)) if len(f) else 0.0
    it = 0
    while not norm_f < tol and it < max_it:
        it += 1
        ds_dvm, ds_dva = _ds_dv(ybus, v)
        j11 = ds_dva[pvpq][:, pvpq].real
        j12 = ds_dvm[pvpq][:, pq].real
        j21 = ds_dva[pq][:, pvpq].imag
        j22 = ds_dvm[pq][:, pq].imag
        jac = sp.vstack([sp.hstack([j11, j12]), sp.hstack([j21, j22])]).tocsc()
        try:
            dx = -splu(jac).solve(f)
        except RuntimeError:            # exactly singular Jacobian
            return v, False, it, np.inf
        va[pvpq] += dx[:npv + npq]
        vm[pq] += dx[npv + npq:]
        v = vm * np.exp(1j * va)
        vm, va = np.abs(v), np.angle(v)
        f = mismatch(v)
        norm_f = np.max(np.abs(f)) if len(f) else 0.0
        if not np.isfinite(norm_f):
            return v, False, it, norm_f
    return v, bool(norm_f < tol), it, float(norm_f)


# --------------------------------------------------------------------------
# P5 + P6: outer q-limit loop and results
# --------------------------------------------------------------------------
def solve_case(case, p_inj, q_inj, qg_min=None, qg_max=None, qd_bus=None,
               enforce_q_lims=False, tol=1e-8, max_it=10, br_status=None, v_init=None):
    """Solve one instance.

    p_inj/q_inj: net bus injections in p.u. (generation − demand); q_inj at PV
    buses is ignored while the bus is PV.  qg_min/qg_max/qd_bus [nb] in p.u.
    describe the reactive capability of the generators at PV buses and the
    reactive demand there (for the enforce_q_lims loop): when the
    solved generator Q leaves [qg_min, qg_max] the bus becomes PQ with Q pinned
    at the violated limit and the case is solved again.
    Returns dict(V, converged, iterations, max_mismatch, bus_type).
    """
    ybus = make_ybus(case, br_status)
    bus_type = case.bus_type.copy()
    p = np.asarray(p_inj, float).copy()
    q = np.asarray(q_inj, float).copy()
    v = case.vm_set * np.exp(1j * case.va_set) if v_init is None else v_init.copy()
    # pandapower `check_connectivity=True`: buses without a path to a slack are taken
    # out of service; their voltages come back as NaN.
    alive = energised_buses(case, br_status)
    total_it = 0
    while True:
        pv = np.flatnonzero((bus_type == PV) & alive)
        pq = np.flatnonzero((bus_type == PQ) & alive)
        v, ok, it, nrm = newtonpf(ybus, p + 1j * q, v, pv, pq, tol, max_it)
        total_it += it
        if not ok or not enforce_q_lims or qg_min is None or len(pv) == 0:
            break
        s_calc = v * np.conj(ybus @ v)
        qg = s_calc.imag[pv] + qd_bus[pv]
        hi = qg > qg_max[pv]
        lo = qg < qg_min[pv]
        if not (hi.any() or lo.any()):
            break
        fix = pv[hi | lo]
        q[pv[hi]] = qg_max[pv[hi]] - qd_bus[pv[hi]]
        q[pv[lo]] = qg_min[pv[lo]] - qd_bus[pv[lo]]
        bus_type[fix] = PQ
    v = np.where(alive, v, np.nan + 0j)
    return dict(V=v, converged=ok, iterations=total_it, max_mismatch=nrm,
                bus_type=bus_type, ybus=ybus)


def energised_buses(case, br_status=None):
    """Buses with a path of in-service branches to a REF bus."""
    nb = case.nb
    on = np.ones(case.nbr, bool) if br_status is None else np.asarray(br_status) != 0
    adj = [[] for _ in range(nb)]
    for k in np.flatnonzero(on):
        adj[int(case.f[k])].append(int(case.t[k])); adj[int(case.t[k])].append(int(case.f[k]))
    alive = np.zeros(nb, bool)
    stack = [int(i) for i in np.flatnonzero(case.bus_type == REF)]
    for i in stack:
        alive[i] = True
    while stack:
        u = stack.pop()
        for w in adj[u]:
            if not alive[w]:
                alive[w] = True
                stack.append(w)
    return alive


def branch_results(case, v, br_status=None):
    """pfsoln branch part + pandapower loading definitions:
    i_ka = |S|/(√3·vm·vn) so |I| p.u. × kf/kt gives percent of rating."""
    s = np.ones(case.nbr) if br_status is None else np.asarray(br_status, float)
    i_f = (case.yff * v[case.f] + case.yft * v[case.t]) * s
    i_t = (case.ytf * v[case.f] + case.ytt * v[case.t]) * s
    s_f = v[case.f] * np.conj(i_f)
    s_t = v[case.t] * np.conj(i_t)
    loading = np.maximum(np.abs(i_f) * case.kf, np.abs(i_t) * case.kt)
    loading = np.where(s == 0, 0.0, loading)              # out of service: 0 %, also next to a dead bus
    return dict(s_from=s_f, s_to=s_t, i_from=np.abs(i_f), i_to=np.abs(i_t),
                loading_percent=loading)


# --------------------------------------------------------------------------
# DataFrame level: the `power_flow_solver(net)` contract (opf_env.py:53,657)
# --------------------------------------------------------------------------
def bus_injections(net, case):
    """makeSbus on the element tables: generation − demand per case bus, MW."""
    nb = case.nb
    p = np.zeros(nb)
    q = np.zeros(nb)
    qd = np.zeros(nb)
    for tbl, sign in (('load', -1.0), ('sgen', 1.0), ('storage', -1.0)):
        df = net[tbl]
        if not len(df):
            continue
        on = df['in_service'].to_numpy(bool) if 'in_service' in df.columns else np.ones(len(df), bool)
        sc = df['scaling'].to_numpy(float) if 'scaling' in df.columns else np.ones(len(df))
        for pos, b in enumerate(df['bus'].to_numpy()):
            if on[pos] and int(b) in case.bus_lookup:
                i = case.bus_lookup[int(b)]
                p[i] += sign * float(df['p_mw'].iloc[pos]) * sc[pos]
                qv = sign * float(df['q_mvar'].iloc[pos]) * sc[pos]
                q[i] += qv
                qd[i] -= qv
    gen = net['gen']
    qmin = np.full(nb, -np.inf)
    qmax = np.full(nb, np.inf)
    if len(gen):
        on = gen['in_service'].to_numpy(bool)
        sc = gen['scaling'].to_numpy(float) if 'scaling' in gen.columns else np.ones(len(gen))
        lim_lo = np.zeros(nb)
        lim_hi = np.zeros(nb)
        has = np.zeros(nb, bool)
        for pos, b in enumerate(gen['bus'].to_numpy()):
            if on[pos] and int(b) in case.bus_lookup:
                i = case.bus_lookup[int(b)]
                p[i] += float(gen['p_mw'].iloc[pos]) * sc[pos]
                lo = float(gen['min_q_mvar'].iloc[pos]) if 'min_q_mvar' in gen.columns else np.nan
                hi = float(gen['max_q_mvar'].iloc[pos]) if 'max_q_mvar' in gen.columns else np.nan
                lim_lo[i] += -np.inf if np.isnan(lo) else lo
                lim_hi[i] += np.inf if np.isnan(hi) else hi
                has[i] = True
        qmin[has], qmax[has] = lim_lo[has], lim_hi[has]
    return p, q, qd, qmin, qmax


def runpp(net, enforce_q_lims=True, tol=1e-8, max_it=10, **kwargs):
    """`power_flow_solver(net)` built on the oracle: same contract as
    OpfEnv.default_power_flow (opf_env.py:696-709) — mutates `net.res_*` in
    place, raises LoadflowNotConverged on failure."""
    import pandas as pd
    from opfgym_amd.case import net_to_case
    case = net_to_case(net)
    base = case.base_mva
    p, q, qd, qmin, qmax = bus_injections(net, case)
    sol = solve_case(case, p / base, q / base, qmin / base, qmax / base, qd / base,
                     enforce_q_lims=enforce_q_lims, tol=tol, max_it=max_it)
    if not sol['converged']:
        exc = kwargs.get('not_converged_exception', LoadflowNotConverged)
        raise exc('power flow did not converge')
    write_results(net, case, sol, p, q, qd)
    return sol


def write_results(net, case, sol, p_mw_bus, q_mvar_bus, qd_mvar_bus):
    import pandas as pd
    v = sol['V']
    base = case.base_mva
    vm = np.full(len(net['bus']), np.nan)
    va = np.full(len(net['bus']), np.nan)
    for pos, b in enumerate(net['bus'].index):
        if int(b) in case.bus_lookup:
            i = case.bus_lookup[int(b)]
            vm[pos], va[pos] = abs(v[i]), np.degrees(np.angle(v[i]))
    net['res_bus'] = pd.DataFrame({'vm_pu': vm, 'va_degree': va}, index=net['bus'].index)
    br = branch_results(case, v)
    for tbl, kind in (('line', 0), ('trafo', 1)):
        load = np.full(len(net[tbl]), np.nan)
        sel = case.br_kind == kind
        load[case.br_elem[sel]] = br['loading_percent'][sel]
        if len(net[tbl]):
            # an element that is out of service or behind an open switch carries no flow: 0 % (pypower's
            # pfsoln zeroes the flows of status-0 branches); an in-service element at a de-energised bus: NaN
            off = ~net[tbl]['in_service'].to_numpy(bool)
            sw = net['switch'] if 'switch' in net else None
            if sw is not None and len(sw):        # behind an open switch: taken out by net_to_case
                opened = sw['element'][(sw['et'] == tbl[0]) & ~sw['closed'].to_numpy(bool)].to_numpy()
                off = off | np.isin(net[tbl].index.to_numpy(), opened)
            load[off & np.isnan(load)] = 0.0
        net['res_' + tbl] = pd.DataFrame({'loading_percent': load}, index=net[tbl].index)
    s_bus = v * np.conj(sol['ybus'] @ v) * base
    eg = net['ext_grid']
    pe = np.full(len(eg), np.nan)
    qe = np.full(len(eg), np.nan)
    for pos, b in enumerate(eg['bus'].to_numpy()):
        if int(b) in case.bus_lookup:
            i = case.bus_lookup[int(b)]
            pe[pos] = s_bus[i].real - p_mw_bus[i]
            qe[pos] = s_bus[i].imag - q_mvar_bus[i]
    net['res_ext_grid'] = pd.DataFrame({'p_mw': pe, 'q_mvar': qe}, index=eg.index)
    for tbl in ('load', 'sgen', 'storage'):
        df = net[tbl]
        sc = df['scaling'].to_numpy(float) if 'scaling' in df.columns and len(df) else 1.0
        net['res_' + tbl] = pd.DataFrame(
            {'p_mw': df['p_mw'].to_numpy(float) * sc if len(df) else [],
             'q_mvar': df['q_mvar'].to_numpy(float) * sc if len(df) else []}, index=df.index)
    gen = net['gen']
    if len(gen):
        sc = gen['scaling'].to_numpy(float) if 'scaling' in gen.columns else 1.0
        qg = np.full(len(gen), np.nan)
        vmg = np.full(len(gen), np.nan)
        for pos, b in enumerate(gen['bus'].to_numpy()):
            if int(b) in case.bus_lookup:
                i = case.bus_lookup[int(b)]
                # reactive output = calculated bus injection + local reactive demand
                qg[pos] = s_bus[i].imag + qd_mvar_bus[i]
                vmg[pos] = abs(v[i])
        net['res_gen'] = pd.DataFrame({'p_mw': gen['p_mw'].to_numpy(float) * sc,
                                       'q_mvar': qg, 'vm_pu': vmg}, index=gen.index)
    else:
        net['res_gen'] = pd.DataFrame({'p_mw': [], 'q_mvar': [], 'vm_pu': []})
