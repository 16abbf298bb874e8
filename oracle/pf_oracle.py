"""CPU oracle for the solver half of the hot path (SURVEY.md §8a rows P2–P7).

TEST INFRASTRUCTURE ONLY.  Nothing in the product package may
import or call this module; only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` do, and only as the checker / the timed CPU
baseline.  Conversely nothing under `oracle/` imports the product: the case
this solver works on is built by `oracle/pd2ppc.py` from the net's tables.

What it restates: the Newton-Raphson AC power flow that the reference reaches
through `pandapower.runpp(net, enforce_q_lims=True)` at
`/root/reference/opfgym/opf_env.py:703`.  pandapower (`>=2.13.1,<3.0`,
pyproject.toml:32) is a third-party dependency that is NOT vendored in
/root/reference and NOT installed here, so this file restates the published
pypower/MATPOWER algorithm it derives from: `makeYbus` (from r, x, b, tap,
shift — not from admittance stamps), `makeSbus`, `newtonpf` (polar full Newton
on [ΔP(pv∪pq); ΔQ(pq)], ∞-norm stop at `tol`, ≤10 iterations, sparse direct
solve — here SciPy SuperLU with partial pivoting, the same class of solver
pandapower uses without lightsim2grid), the per-generator `enforce_q_lims`
outer loop of pypower's `runpf` (a generator beyond a limit is switched off,
its bus becomes PQ and carries the limit as negative demand), `pfsoln` (branch
flows; reactive dispatch split among generators sharing a bus in proportion to
their ranges) and pandapower's result definitions (`i_ka`, `loading_percent`
of lines and transformers with `trafo_loading='current'`).

PARITY against pandapower: pinned by PUBLISHED pandapower numbers, not by a pandapower run.  The
reference's own tests hold no numerical power-flow result at all and pandapower cannot
be run here, so `fixtures/` (exports written by scripts/export_pandapower_case.py wherever pandapower
exists) is empty.  What pins this oracle (tests/test_oracle_pf.py, tests/test_pandapower_published.py):
  * constants pandapower itself publishes [3P, from memory, kept only where every digit reproduced —
    tests/pandapower_published.py]: the documentation's minimal example (res_bus / res_ext_grid to the six
    printed decimals) and six networks of pandapower's own test/loadflow/test_results.py ("result
    values from powerfactory", asserted there and here at 1e-6 p.u. / 1e-3 % / 1e-6 kA):
      element formula                                            exercised by
      line pi-model, parallel, df, loading_percent / i_ka        tests_line, tests_load_sgen
      line open at one end (auxiliary bus), line out of service  tests_line
      load / sgen injections, unit in_service mask               tests_load_sgen
      2w transformer from vk/vkr/pfe/i0 (T-model), lv cable      docs_minimal_example
      2w transformer tap (hv), parallel, open at one side,       tests_trafo (hv bus voltage and loading only)
        loading_percent with trafo_loading='current'
      3w transformer star equivalent, hv tap, loading            tests_trafo3w
      PV bus, generator reactive dispatch                        tests_gen
      enforce_q_lims (PV -> PQ at the limit)                     tests_enforce_qlims
      slack P/Q (res_ext_grid)                                   docs_minimal_example
    (These vectors are recalled from memory and were kept only where the oracle reproduced every digit: read them as
    self-consistency pins, see the header of tests/pandapower_published.py; the ones that did not reproduce are listed there
    and asserted as expected failures.)
    Exercised by no recalled number, pinned by EQUIVALENCE instead (tests/metamorphic.py: two formulations that pandapower's
    model definitions make the same problem must give the same answer — on this oracle, tests/test_metamorphic.py, and on
    the GPU, tests/test_gpu_metamorphic.py):
      closed bus-bus switch = one bus; open one = none          bus_bus_switch_is_one_bus, open_bus_bus_switch_is_no_switch
      tap on the lv side / hv side = a changed rated voltage    lv_side_tap_is_a_changed_lv_rating, hv_side_tap_...
      parallel = 2 (line, transformer) = two elements           parallel_two_is_two_elements
      vector-group shift under calculate_voltage_angles         vector_group_shift_turns_the_angles_behind_it (144-bus grid:
                                                                 no |V| / loading / slack power moves, angles turn by 150 deg)
      shunt (p, q, step) = a load of p |V|^2, q |V|^2           shunt_is_a_constant_impedance_load
      several ext_grids                                          two_ext_grids_at_one_set_point_are_a_fused_slack
      ideal phase shifter (tap_phase_shifter, tap_step_degree)   ideal_phase_shifter_is_a_changed_vector_group (in a loop with a plain
                                                                 transformer: the circulating flow; hv and lv side)
      storages (sign convention, scaling)                        storage_is_a_load
      ward (constant power + constant impedance)                ward_is_a_load_and_a_shunt
      xward (ward + internal source behind an impedance)         xward_is_a_ward_and_a_voltage_source_behind_an_impedance
      dcline (two generators, losses, reactive ranges)           dcline_is_two_generators
      motor (pn_mech, efficiency, loading, cos_phi)              motor_is_a_load
      series impedance, both directions alike                    symmetric_impedance_is_a_line_without_charging (also 3 r I^2 losses)
      series impedance, directions differing                     tests/test_beyond_simbench_elements.py (by hand: each side's own equation)
      closed bus-bus switch with z_ohm (switch_rx_ratio = 2)     bus_bus_switch_with_impedance_is_a_short_line
      generators sharing a bus (pfsoln's split)                 tests/test_generator_dispatch.py (by hand: shares of the ranges)
      `_is_elements` zero rule of result rows                   tests/test_gpu_env.py::test_units_on_a_de_energised_island_cost_nothing
      DC start values                                            the converged solution does not depend on them; iteration counts
                                                                 against this oracle's own DC start only
      line conductance g_us_per_km (half at each end)            line_conductance_is_a_shunt_at_each_end
      derating factors df of lines and transformers             derating_factor_scales_the_loading
      ideal phase shifter given as tap_step_percent              phase_shifter_given_in_percent_is_one_given_in_degrees (2 asin(du / 2))
    With these, every formula of the converter is exercised by a recalled number or by an equivalence.
  * the closed-form two-bus solution,
  * published load-flow solutions of textbook systems: WSCC 9-bus (Anderson & Fouad), IEEE 14-bus
    (off-nominal taps, bus shunt, four PV buses; |V| to the three published decimals, angles to
    0.001 degree, slack generation and losses to 0.01 MW), IEEE 30-bus, Grainger & Stevenson 4-bus,
    Wood & Wollenberg 6-bus, Stagg & El-Abiad 5-bus,
  * algebraic self-checks (mismatch < tol, power balance = losses).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
from scipy.sparse.linalg import splu

from . import pd2ppc
from .pd2ppc import NONE, PPC, PQ, PV, REF

EPS = np.finfo(float).eps


class LoadflowNotConverged(Exception):
    """Stand-in for pandapower.powerflow.LoadflowNotConverged (opf_env.py:660)."""


# --------------------------------------------------------------------------
# P3: makeYbus / makeSbus
# --------------------------------------------------------------------------
def branch_admittances(ppc: PPC, status=None):
    """pypower makeYbus, branch part: Ys = stat/(r+jx); Bc = stat*b; tap = ratio e^{j shift};
    Ytt = Ys + j Bc/2; Yff = Ytt/(tap conj(tap)); Yft = -Ys/conj(tap); Ytf = -Ys/tap."""
    stat = (ppc.status if status is None else np.asarray(status)).astype(float)
    ys = stat / (ppc.r + 1j * ppc.x)
    bc = stat * ppc.b
    tap = ppc.tap * np.exp(1j * np.pi / 180.0 * ppc.shift)
    # pandapower's makeYbus (`branch_vectors`): the to side of a branch with BR_R_ASYM / BR_X_ASYM sees its own series
    # admittance (net.impedance with rtf_pu != rft_pu)
    ys_t = ys if ppc.r_asym is None else stat / ((ppc.r + ppc.r_asym) + 1j * (ppc.x + ppc.x_asym))
    ytt = ys_t + 1j * bc / 2.0
    yff = (ys + 1j * bc / 2.0) / (tap * np.conj(tap))
    yft = -ys / np.conj(tap)
    ytf = -ys_t / tap
    return yff, yft, ytf, ytt


def make_ybus(ppc: PPC, status=None) -> sp.csr_matrix:
    nb, nl = ppc.nb, ppc.nbr
    yff, yft, ytf, ytt = branch_admittances(ppc, status)
    ysh = (ppc.gs + 1j * ppc.bs) / ppc.base_mva
    cf = sp.csr_matrix((np.ones(nl), (np.arange(nl), ppc.f)), (nl, nb))
    ct = sp.csr_matrix((np.ones(nl), (np.arange(nl), ppc.t)), (nl, nb))
    yf = sp.diags(yff) @ cf + sp.diags(yft) @ ct
    yt = sp.diags(ytf) @ cf + sp.diags(ytt) @ ct
    return (cf.T @ yf + ct.T @ yt + sp.diags(ysh)).tocsr()


def make_sbus(ppc: PPC, pd=None, qd=None, g_p=None, g_q=None, g_status=None):
    """Net complex bus injection in p.u.: generation - demand."""
    pd = ppc.pd if pd is None else pd
    qd = ppc.qd if qd is None else qd
    on = (ppc.g_status if g_status is None else g_status) > 0
    s = -(pd + 1j * qd)
    gp = ppc.g_p if g_p is None else g_p
    gq = np.zeros(len(ppc.g_bus)) if g_q is None else g_q
    np.add.at(s, ppc.g_bus[on], gp[on] + 1j * gq[on])
    return s / ppc.base_mva


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


def start_voltage(ppc: PPC, init='flat'):
    """'flat': |V| = 1 (set-points at PV/REF buses), angle = the slack angle carried through the
    transformer phase shifts (a 150 degree vector group makes a literal flat start useless; pandapower's
    default for such grids is init='dc').  'dc': angles of the DC power flow (pandapower `init='dc'`)."""
    nb = ppc.nb
    va = np.radians(ppc.va.copy())
    seen = ppc.bus_type == REF
    adj = [[] for _ in range(nb)]
    for k in range(ppc.nbr):
        if ppc.status[k]:
            sh = np.radians(ppc.shift[k])
            adj[int(ppc.f[k])].append((int(ppc.t[k]), -sh))
            adj[int(ppc.t[k])].append((int(ppc.f[k]), +sh))
    seen = seen.copy()
    stack = [int(i) for i in np.flatnonzero(seen)]
    while stack:
        a = stack.pop()
        for w, d in adj[a]:
            if not seen[w]:
                seen[w] = True
                va[w] = va[a] + d
                stack.append(w)
    if init == 'dc':
        va = _dc_angles(ppc, va)
    return ppc.vm * np.exp(1j * va)


def _dc_angles(ppc: PPC, va_start):
    """pypower `makeBdc` + `dcpf` on the supplied part of the grid."""
    on = ppc.status > 0
    nb = ppc.nb
    bdc = on / ppc.x / ppc.tap
    nl = ppc.nbr
    cft = sp.csr_matrix((np.r_[np.ones(nl), -np.ones(nl)], (np.r_[np.arange(nl), np.arange(nl)], np.r_[ppc.f, ppc.t])), (nl, nb))
    bf = sp.diags(bdc) @ cft
    bbus = (cft.T @ bf).tocsr()
    pfinj = bdc * (-np.radians(ppc.shift))
    pbusinj = cft.T @ pfinj
    s = make_sbus(ppc).real - pbusinj - ppc.gs / ppc.base_mva
    live = ppc.bus_type != NONE
    free = np.flatnonzero(live & (ppc.bus_type != REF))
    ref = np.flatnonzero(ppc.bus_type == REF)
    va = va_start.copy()
    if len(free):
        rhs = s[free] - bbus[free][:, ref] @ va[ref]
        va[free] = splu(bbus[free][:, free].tocsc()).solve(rhs)
    return va


# --------------------------------------------------------------------------
# P5 + P6: outer q-limit loop (pypower runpf) and pfsoln
# --------------------------------------------------------------------------
def _gen_q_dispatch(ppc: PPC, sbus_calc, qd, g_status):
    """pypower `pfsoln`, generator part: total reactive injection of a bus + local demand, shared
    among the generators of the bus in proportion to their reactive ranges."""
    on = np.flatnonzero(g_status > 0)
    qg = np.zeros(len(ppc.g_bus))
    if not len(on):
        return qg
    gbus = ppc.g_bus[on]
    total = sbus_calc.imag[gbus] * ppc.base_mva + qd[gbus]
    n_at = np.zeros(ppc.nb)
    np.add.at(n_at, gbus, 1.0)
    qg[on] = total / n_at[gbus]
    if len(on) > 1:
        qmin_b, qmax_b = np.zeros(ppc.nb), np.zeros(ppc.nb)
        np.add.at(qmin_b, gbus, ppc.g_qmin[on])
        np.add.at(qmax_b, gbus, ppc.g_qmax[on])
        q_tot = np.zeros(ppc.nb)
        np.add.at(q_tot, gbus, qg[on])
        zero_range = qmin_b[gbus] == qmax_b[gbus]
        share = ppc.g_qmin[on] + ((q_tot - qmin_b) / (qmax_b - qmin_b + EPS))[gbus] * (ppc.g_qmax[on] - ppc.g_qmin[on])
        qg[on] = np.where(zero_range, qg[on], share)
    return qg


def solve(ppc: PPC, enforce_q_lims=False, tol=1e-8, max_it=10, status=None, v_init=None, init='flat'):
    """Solve the case.  `status`: optional branch status vector replacing ppc.status (N-1 / outage
    studies); buses that lose their supply are isolated as pandapower's `check_connectivity` does.
    Returns dict(V, converged, iterations, max_mismatch, bus_type, ybus, qg, pg, supplied)."""
    st = ppc.status if status is None else np.asarray(status).astype(np.int64)
    ybus = make_ybus(ppc, st)
    supplied = pd2ppc.supplied_buses(ppc, st) & (ppc.bus_type != NONE)
    bus_type = np.where(supplied, ppc.bus_type, NONE)
    g_status = ppc.g_status * supplied[ppc.g_bus]
    pd_, qd_ = ppc.pd.copy(), ppc.qd.copy()
    if v_init is not None:
        v = v_init.copy()
    else:
        # (the start is computed on the grid as it is solved: branch status of this call, unsupplied buses taken out — a DC
        #  start on a grid with a cut-off island would otherwise be singular; pandapower drops such buses before its DC run)
        saved = ppc.status, ppc.bus_type
        ppc.status, ppc.bus_type = st, bus_type
        try:
            v = start_voltage(ppc, init)
        finally:
            ppc.status, ppc.bus_type = saved
    v = np.where(supplied, v, 1.0 + 0j)
    total_it = 0
    g_q = np.zeros(len(ppc.g_bus))
    fixed = np.zeros(len(ppc.g_bus), dtype=bool)        # generators pinned at a reactive limit
    while True:
        # a bus keeps its PV type only while a voltage-controlling generator is on there (pypower `bustypes`)
        has_gen = np.zeros(ppc.nb, dtype=bool)
        has_gen[ppc.g_bus[(g_status > 0) & ~fixed]] = True
        bt = np.where((bus_type == PV) & ~has_gen, PQ, bus_type)
        pv = np.flatnonzero(bt == PV)
        pq = np.flatnonzero(bt == PQ)
        free_status = np.where(fixed, 0, g_status)
        sbus = make_sbus(ppc, pd_, qd_, g_status=free_status)
        v, ok, it, nrm = newtonpf(ybus, sbus, v, pv, pq, tol, max_it)
        total_it += it
        s_calc = v * np.conj(ybus @ v)
        qg_free = _gen_q_dispatch(ppc, s_calc, qd_, free_status)
        g_q = np.where(fixed, g_q, qg_free)
        if not ok or not enforce_q_lims:
            break
        cand = (free_status > 0) & (bt[ppc.g_bus] == PV)
        hi = cand & (g_q > ppc.g_qmax)
        lo = cand & (g_q < ppc.g_qmin)
        if not (hi.any() or lo.any()):
            break
        g_q = np.where(hi, ppc.g_qmax, np.where(lo, ppc.g_qmin, g_q))
        for g in np.flatnonzero(hi | lo):               # switched off, output booked as negative demand
            fixed[g] = True
            pd_[ppc.g_bus[g]] -= ppc.g_p[g]
            qd_[ppc.g_bus[g]] -= g_q[g]
    # slack generation (pfsoln): P of the first generator at each REF bus balances the bus
    g_p = ppc.g_p.copy()
    for i in np.flatnonzero(bus_type == REF):
        at = np.flatnonzero((ppc.g_bus == i) & (g_status > 0))
        if len(at):
            others = g_p[at[1:]].sum()
            g_p[at[0]] = s_calc[i].real * ppc.base_mva + ppc.pd[i] - others
    v_out = np.where(supplied, v, np.nan + 0j)
    return dict(V=v_out, converged=ok, iterations=total_it, max_mismatch=nrm, bus_type=bt, ybus=ybus,
                qg=g_q, pg=g_p, supplied=supplied, status=st, fixed=fixed)


def branch_flows(ppc: PPC, v, status=None):
    """pfsoln, branch part: complex power into the branch at both ends, p.u."""
    st = ppc.status if status is None else np.asarray(status)
    yff, yft, ytf, ytt = branch_admittances(ppc, st)
    vf, vt = v[ppc.f], v[ppc.t]
    s_f = vf * np.conj(yff * vf + yft * vt)
    s_t = vt * np.conj(ytf * vf + ytt * vt)
    return s_f, s_t


def loading_percent(ppc: PPC, net, v, status=None):
    """pandapower `_get_line_results` / `_get_trafo_results` / `_get_trafo3w_results`:
    i_ka = |S| / (sqrt(3) |V| vn_bus); lines: max(i_from, i_to) / (max_i_ka df parallel);
    transformers (`trafo_loading='current'`): max over the windings of i_ka vn_rated sqrt(3) / sn,
    divided by parallel and df.  A branch that is switched off carries no power: 0 %; |V| of an
    isolated bus is NaN and so is every current computed with it (the maxima propagate NaN, as
    numpy's do).  Returns {table: array over the table's rows}."""
    st = ppc.status if status is None else np.asarray(status)
    s_f, s_t = branch_flows(ppc, np.nan_to_num(v, nan=0.0), st)
    vm = np.abs(v)
    with np.errstate(divide='ignore', invalid='ignore'):
        i_f = np.abs(s_f) * ppc.base_mva / (np.sqrt(3.0) * vm[ppc.f] * ppc.base_kv[ppc.f])
        i_t = np.abs(s_t) * ppc.base_mva / (np.sqrt(3.0) * vm[ppc.t] * ppc.base_kv[ppc.t])
    out = {}
    for table in ('line', 'trafo', 'trafo3w'):
        df = net[table] if table in net else None
        n = 0 if df is None else len(df)
        res = np.full(n, np.nan)
        first = np.ones(n, dtype=bool)
        for k in np.flatnonzero(np.array([tb == table for tb in ppc.br_table], dtype=bool)) if n else ():
            pos = int(ppc.br_pos[k])
            if table == 'line':
                i_max = float(df['max_i_ka'].iloc[pos]) * _opt(df, 'df', pos, 1.0) * _opt(df, 'parallel', pos, 1.0)
                res[pos] = np.maximum(i_f[k], i_t[k]) / i_max * 100.0
            elif table == 'trafo':
                sn = float(df['sn_mva'].iloc[pos])
                ld = np.maximum(i_f[k] * float(df['vn_hv_kv'].iloc[pos]), i_t[k] * float(df['vn_lv_kv'].iloc[pos]))
                res[pos] = ld * np.sqrt(3.0) / sn * 100.0 / _opt(df, 'parallel', pos, 1.0) / _opt(df, 'df', pos, 1.0)
            else:
                side = ppc.br_side[k]
                sn = float(df[f'sn_{side}_mva'].iloc[pos])
                i_side = i_f[k] if side == 'hv' else i_t[k]          # current at the terminal, not at the star point
                ld = i_side * float(df[f'vn_{side}_kv'].iloc[pos]) * np.sqrt(3.0) / sn * 100.0
                res[pos] = ld if first[pos] else np.maximum(res[pos], ld)
                first[pos] = False
        out[table] = res
    return out


def _opt(df, col, pos, default):
    if col not in df.columns:
        return default
    v = df[col].iloc[pos]
    try:
        v = float(v)
    except (TypeError, ValueError):
        return default
    return default if np.isnan(v) else v


# --------------------------------------------------------------------------
# DataFrame level: the `power_flow_solver(net)` contract (opf_env.py:53,657)
# --------------------------------------------------------------------------
def runpp(net, enforce_q_lims=True, tol=1e-8, max_it=10, init='flat', **kwargs):
    """`power_flow_solver(net)` built on the oracle: same contract as
    OpfEnv.default_power_flow (opf_env.py:696-709) — mutates `net.res_*` in
    place, raises LoadflowNotConverged on failure."""
    ppc = pd2ppc.build_ppc(net)
    sol = solve(ppc, enforce_q_lims=enforce_q_lims, tol=tol, max_it=max_it, init=init)
    if not sol['converged']:
        exc = kwargs.get('not_converged_exception', LoadflowNotConverged)
        raise exc('power flow did not converge')
    write_results(net, ppc, sol)
    sol['ppc'] = ppc
    return sol


def write_results(net, ppc: PPC, sol):
    import pandas as pd
    v = sol['V']
    vm = np.full(len(net['bus']), np.nan)
    va = np.full(len(net['bus']), np.nan)
    on = net['bus']['in_service'].to_numpy(bool) if 'in_service' in net['bus'].columns else np.ones(len(vm), bool)
    for pos, b in enumerate(net['bus'].index):
        i = ppc.bus_lookup[int(b)]
        if on[pos] and sol['supplied'][i]:
            vm[pos], va[pos] = abs(v[i]), np.degrees(np.angle(v[i]))
    net['res_bus'] = pd.DataFrame({'vm_pu': vm, 'va_degree': va}, index=net['bus'].index)
    ld = loading_percent(ppc, net, v, sol['status'])
    for table in ('line', 'trafo', 'trafo3w'):
        if table in net:
            net['res_' + table] = pd.DataFrame({'loading_percent': ld[table]}, index=net[table].index)
    # Units that take no part in the power flow — out of service themselves, or on a bus that is out of service or
    # cut off every slack (pandapower's `_is_elements`: `in_service` AND the bus in service, isolated buses counted as
    # out of service) — get ZEROS in their result rows: pandapower fills `res_gen` from arrays of zeros for the
    # in-service generators only (results_gen.py: p, q, vm_pu, va_degree) and multiplies the set-points of
    # `res_load / res_sgen / res_storage` by that mask (results_bus.py: write_pq_results_to_element).  [3P, from the
    # published source, unverified here.]
    eg, gen = net['ext_grid'], net['gen']
    pe, qe = np.full(len(eg), np.nan), np.full(len(eg), np.nan)
    pg, qg, vg = np.zeros(len(gen)), np.zeros(len(gen)), np.zeros(len(gen))
    for g in range(len(ppc.g_bus)):
        live = ppc.g_status[g] > 0 and sol['supplied'][ppc.g_bus[g]]
        pos = int(ppc.g_pos[g])
        if ppc.g_table[g] == 'ext_grid':
            if live:
                pe[pos], qe[pos] = sol['pg'][g], sol['qg'][g]
        elif live and ppc.g_table[g] == 'gen':       # (the internal sources of xwards: res_xward, below)
            pg[pos], qg[pos], vg[pos] = sol['pg'][g], sol['qg'][g], abs(v[ppc.g_bus[g]])
    net['res_ext_grid'] = pd.DataFrame({'p_mw': pe, 'q_mvar': qe}, index=eg.index)
    if len(gen):
        net['res_gen'] = pd.DataFrame({'p_mw': pg, 'q_mvar': qg, 'vm_pu': vg}, index=gen.index)
    else:
        net['res_gen'] = pd.DataFrame({'p_mw': [], 'q_mvar': [], 'vm_pu': []})
    bus_alive = {int(b): bool(on[pos] and sol['supplied'][ppc.bus_lookup[int(b)]]) for pos, b in enumerate(net['bus'].index)}
    for tbl in ('load', 'sgen', 'storage'):
        df = net[tbl]
        sc = df['scaling'].to_numpy(float) if 'scaling' in df.columns and len(df) else 1.0
        part = np.array([bus_alive.get(int(b), False) for b in df['bus']], dtype=float) if len(df) else 1.0
        if len(df) and 'in_service' in df.columns:
            part = part * df['in_service'].to_numpy(bool)
        net['res_' + tbl] = pd.DataFrame(
            {'p_mw': df['p_mw'].to_numpy(float) * sc * part if len(df) else [],
             'q_mvar': df['q_mvar'].to_numpy(float) * sc * part if len(df) else []}, index=df.index)
    # ---- element types beyond the SimBench grids (pandapower results_bus.py / results_branch.py) ------------------------
    def alive_at(df):
        part = np.array([bus_alive.get(int(b), False) for b in df['bus']], dtype=bool)
        return part & (df['in_service'].to_numpy(bool) if 'in_service' in df.columns else True)
    if 'ward' in net and len(net['ward']):
        df = net['ward']
        part = alive_at(df)
        vmw = np.array([vm[net['bus'].index.get_loc(int(b))] for b in df['bus']])      # (the bus's voltage, in service or not)
        v2 = np.where(part, vmw, 0.0) ** 2
        net['res_ward'] = pd.DataFrame({'p_mw': part * (df['ps_mw'].to_numpy(float) + df['pz_mw'].to_numpy(float) * v2),
                                        'q_mvar': part * (df['qs_mvar'].to_numpy(float) + df['qz_mvar'].to_numpy(float) * v2),
                                        'vm_pu': vmw}, index=df.index)
    if 'dcline' in net and len(net['dcline']):
        # pandapower `_get_dcline_results`: the two generators of a line, seen from the line (consumption positive)
        df = net['dcline']
        cols = {c: np.zeros(len(df)) for c in ('p_from_mw', 'q_from_mvar', 'p_to_mw', 'q_to_mvar', 'pl_mw')}
        for g in range(len(ppc.g_bus)):
            if ppc.g_table[g] == 'dcline' and ppc.g_status[g] > 0 and sol['supplied'][ppc.g_bus[g]]:
                row, end = divmod(int(ppc.g_pos[g]), 2)
                side = 'to' if end == 0 else 'from'
                cols[f'p_{side}_mw'][row], cols[f'q_{side}_mvar'][row] = -sol['pg'][g], -sol['qg'][g]
        cols['pl_mw'] = cols['p_from_mw'] + cols['p_to_mw']
        for side in ('from', 'to'):
            at = [net['bus'].index.get_loc(int(b)) for b in df[side + '_bus']]
            cols[f'vm_{side}_pu'], cols[f'va_{side}_degree'] = vm[at], va[at]
        net['res_dcline'] = pd.DataFrame(cols, index=df.index)
    if 'xward' in net and len(net['xward']):
        df = net['xward']
        part = alive_at(df)
        vmw = np.array([vm[net['bus'].index.get_loc(int(b))] for b in df['bus']])
        v2 = np.where(part, vmw, 0.0) ** 2
        px = part * (df['ps_mw'].to_numpy(float) + df['pz_mw'].to_numpy(float) * v2)
        qx = part * (df['qs_mvar'].to_numpy(float) + df['qz_mvar'].to_numpy(float) * v2)
        vi, ai = np.full(len(df), np.nan), np.full(len(df), np.nan)
        s_f, _ = branch_flows(ppc, np.nan_to_num(v, nan=0.0), sol['status'])
        for k in np.flatnonzero(np.array([tb == 'xward' for tb in ppc.br_table], dtype=bool)):
            pos = int(ppc.br_pos[k])
            px[pos] += s_f[k].real * ppc.base_mva           # what flows into the impedance towards the internal source
            qx[pos] += s_f[k].imag * ppc.base_mva
            vi[pos], ai[pos] = abs(v[ppc.t[k]]), np.degrees(np.angle(v[ppc.t[k]]))
        net['res_xward'] = pd.DataFrame({'p_mw': px, 'q_mvar': qx, 'vm_pu': vmw, 'va_internal_degree': ai, 'vm_internal_pu': vi},
                                        index=df.index)
    if 'motor' in net and len(net['motor']):
        df = net['motor']
        p_m, q_m = pd2ppc.motor_pq(df)
        part = alive_at(df).astype(float)
        net['res_motor'] = pd.DataFrame({'p_mw': p_m * part, 'q_mvar': q_m * part}, index=df.index)
    if 'impedance' in net and len(net['impedance']):
        df = net['impedance']
        s_f, s_t = branch_flows(ppc, np.nan_to_num(v, nan=0.0), sol['status'])
        cols = {c: np.zeros(len(df)) for c in ('p_from_mw', 'q_from_mvar', 'p_to_mw', 'q_to_mvar', 'pl_mw', 'ql_mvar', 'i_from_ka', 'i_to_ka')}
        for k in np.flatnonzero(np.array([tb == 'impedance' for tb in ppc.br_table], dtype=bool)):
            pos = int(ppc.br_pos[k])
            sf, st_ = s_f[k] * ppc.base_mva, s_t[k] * ppc.base_mva
            cols['p_from_mw'][pos], cols['q_from_mvar'][pos], cols['p_to_mw'][pos], cols['q_to_mvar'][pos] = sf.real, sf.imag, st_.real, st_.imag
            cols['pl_mw'][pos], cols['ql_mvar'][pos] = (sf + st_).real, (sf + st_).imag
            with np.errstate(divide='ignore', invalid='ignore'):
                cols['i_from_ka'][pos] = abs(sf) / (np.sqrt(3.0) * abs(v[ppc.f[k]]) * ppc.base_kv[ppc.f[k]])
                cols['i_to_ka'][pos] = abs(st_) / (np.sqrt(3.0) * abs(v[ppc.t[k]]) * ppc.base_kv[ppc.t[k]])
        net['res_impedance'] = pd.DataFrame(cols, index=df.index)
