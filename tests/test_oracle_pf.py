"""CPU: pin the power-flow oracle (oracle/pf_oracle.py) with known answers.
pandapower is unavailable, and the reference's tests hold no power-flow result
(SURVEY §8c), so the pins are: the closed-form two-bus solution, the published
WSCC 9-bus load flow, and algebraic self-checks."""
import numpy as np
import pytest

from opfgym_amd import grids
from oracle import pf_oracle as po


@pytest.mark.parametrize('p,q,r,x', [(0.8, 0.3, 0.5, 0.8), (2.0, 1.0, 1.2, 0.9), (0.1, -0.05, 0.2, 2.0)])
def test_two_bus_closed_form(p, q, r, x):
    net = grids.two_bus(p, q, r, x, vn_kv=10.0)
    sol = po.runpp(net)
    assert sol['converged']
    assert abs(net.res_bus.vm_pu.iloc[1] - grids.two_bus_closed_form(p, q, r, x, 10.0)) < 1e-10


def test_wscc9_published_solution():
    net = grids.case9()
    sol = po.runpp(net)
    assert sol['converged'] and sol['iterations'] <= 5
    assert np.abs(net.res_bus.vm_pu.to_numpy() - grids.CASE9_VM).max() < 1e-3
    assert np.abs(net.res_bus.va_degree.to_numpy() - grids.CASE9_VA_DEG).max() < 0.06
    assert abs(net.res_ext_grid.p_mw.iloc[0] - grids.CASE9_SLACK_PQ[0]) < 0.06
    assert abs(net.res_ext_grid.q_mvar.iloc[0] - grids.CASE9_SLACK_PQ[1]) < 0.06
    assert np.abs(net.res_gen.q_mvar.to_numpy() - np.array(grids.CASE9_GEN_Q)).max() < 0.06


@pytest.mark.parametrize('code', ['1-LV-rural1--0-sw', '1-MV-urban--0-sw', 'hv-small'])
def test_self_consistency(code):
    net, _ = grids.get_grid(code)
    sol = po.runpp(net, enforce_q_lims=False)
    ppc, v = sol['ppc'], sol['V']
    s_sched = po.make_sbus(ppc)
    mis = v * np.conj(sol['ybus'] @ v) - s_sched
    assert np.abs(mis.real[ppc.bus_type != 3]).max() < 1e-8
    assert np.abs(mis.imag[ppc.bus_type == 1]).max() < 1e-8
    # power balance: the injections of all buses = branch losses + shunt consumption
    s_f, s_t = po.branch_flows(ppc, v)
    losses = (s_f + s_t).sum() * ppc.base_mva
    s_bus = (v * np.conj(sol['ybus'] @ v)).sum() * ppc.base_mva
    shunt = (np.abs(v) ** 2 * (ppc.gs - 1j * ppc.bs)).sum()
    assert abs(s_bus - losses - shunt) < 1e-6
    assert losses.real > 0
    # the two start-value options of pandapower end in the same solution
    dc = po.solve(ppc, init='dc')
    assert dc['converged'] and np.abs(dc['V'] - v).max() < 1e-9 and dc['iterations'] <= sol['iterations']


def test_q_limits_switch_pv_to_pq():
    net = grids.case9()
    net.gen['min_q_mvar'] = -5.0
    net.gen['max_q_mvar'] = 5.0
    sol = po.runpp(net, enforce_q_lims=True)
    assert sol['converged']
    assert (sol['bus_type'][[1, 2]] == 1).all()            # both generators hit a limit
    assert np.allclose(np.abs(net.res_gen.q_mvar), 5.0, atol=1e-6)
    assert (np.abs(net.res_gen.vm_pu - 1.025) > 1e-4).all()


def test_not_converged_raises():
    net = grids.two_bus(p_mw=500.0, q_mvar=200.0)
    with pytest.raises(po.LoadflowNotConverged):
        po.runpp(net)


def test_ieee14_published_solution():
    """Known-answer test on a second public system with off-nominal taps, a bus shunt and four PV
    buses: the IEEE 14-bus case in pypower matrix form (tests/helpers.ieee14_ppc) against its
    published solution — |V| to the three published decimals, angles to 0.001 degree, slack
    generation 232.39 MW / -16.55 MVAr, losses 13.39 MW."""
    from helpers import ieee14_ppc, oracle_ppc_solve
    base, bus, branch, gen, pub = ieee14_ppc()
    sol = oracle_ppc_solve(base, bus, branch, gen)
    v = sol['V']
    assert sol['converged'] and sol['iterations'] <= 5
    assert np.abs(np.abs(v) - pub['vm']).max() < 6e-4            # published to 3 decimals
    assert np.abs(np.degrees(np.angle(v)) - pub['va_deg']).max() < 1e-3
    s = v * np.conj(sol['ybus'] @ v) * base
    assert abs(sol['pg'][0] - pub['p_slack_mw']) < 0.01
    assert abs(sol['qg'][0] - pub['q_slack_mvar']) < 0.01
    assert abs(s.real.sum() - pub['losses_mw']) < 0.01


def test_ieee30_published_solution():
    """A larger public system: IEEE 30-bus (41 branches, four off-nominal taps, two shunts, five PV buses) against
    numbers of its published load flow (tests/helpers.ieee30_ppc): slack generation, losses, generator reactive
    outputs and a set of bus voltages, each to its printed precision."""
    from helpers import ieee30_ppc, oracle_ppc_solve
    base, bus, branch, gen, pub = ieee30_ppc()
    sol = oracle_ppc_solve(base, bus, branch, gen)
    v = sol['V']
    assert sol['converged'] and sol['iterations'] <= 5
    for b, val in pub['vm'].items():
        assert abs(abs(v[b]) - val) < pub['vm_tol'], ('vm', b, abs(v[b]))
    for b, val in pub['va_deg'].items():
        assert abs(np.degrees(np.angle(v[b])) - val) < pub['va_tol'], ('va', b, np.degrees(np.angle(v[b])))
    assert abs(sol['pg'][0] - pub['p_slack_mw']) < pub['s_tol']
    assert abs(sol['pg'].sum() - bus[:, 2].sum() - pub['losses_mw']) < pub['s_tol']
    for g, val in pub['qg_mvar'].items():
        assert abs(sol['qg'][g] - val) < pub['s_tol'], ('qg', g, sol['qg'][g])


@pytest.mark.parametrize('name', ['gs4', 'ww6', 'sea5'])
def test_published_textbook_solutions(name):
    """Three more public systems whose solved load flow is printed in their textbooks (tests/helpers.py
    `published_cases`): voltages, angles and generator outputs to the printed precision."""
    from helpers import oracle_ppc_solve, published_cases
    base, bus, branch, gen, pub = published_cases()[name]
    sol = oracle_ppc_solve(base, bus, branch, gen)
    v = sol['V']
    assert sol['converged'] and sol['iterations'] <= 5
    assert np.abs(np.abs(v) - np.array(pub['vm'])).max() < pub['vm_tol']
    assert np.abs(np.degrees(np.angle(v)) - np.array(pub['va_deg'])).max() < pub['va_tol']
    for g, val in pub['pg'].items():
        assert abs(sol['pg'][g] - val) < pub['s_tol'], ('pg', g, sol['pg'][g])
    for g, val in pub['qg'].items():
        assert abs(sol['qg'][g] - val) < pub['s_tol'], ('qg', g, sol['qg'][g])


def _fixture_files():
    import glob
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return sorted(glob.glob(os.path.join(root, 'fixtures', '*.npz')))


def _check_export(path):
    """One file of scripts/export_pandapower_case.py: (1) the solver on the exported pypower matrices against the
    exported voltages, (2) the oracle's table converter + solver on the exported element tables against the
    exported result tables (row P2)."""
    import sys, os
    from oracle import pd2ppc
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'scripts'))
    from export_pandapower_case import load_tables
    z = np.load(path, allow_pickle=False)
    ppc = pd2ppc.ppc_from_matrices(float(z['baseMVA']), z['bus'], z['branch'], z['gen'])
    ppc.b = ppc.b - 1j * z['br_g']
    sol = po.solve(ppc, enforce_q_lims=bool(z['enforce_q_lims']))
    assert sol['converged'], path
    assert np.abs(np.abs(sol['V']) - z['res_vm']).max() < 1e-6, path
    assert np.abs(np.degrees(np.angle(sol['V'])) - z['res_va']).max() < 1e-5, path
    net = load_tables(z)
    po.runpp(net, enforce_q_lims=True)
    n_tables = 0
    for key in z.files:
        if key.startswith('out__'):
            _, tbl, col = key.split('__')
            tol = 1e-6 if col in ('vm_pu',) else 1e-4
            assert np.allclose(net[tbl][col].to_numpy(float), z[key], rtol=0, atol=tol, equal_nan=True), (path, key)
            n_tables += 1
    return n_tables


def test_pandapower_export_fixtures():
    """The 1e-6 gate against pandapower proper: every fixtures/*.npz written by
    scripts/export_pandapower_case.py on a machine that has pandapower (element tables, ppci matrices and
    pandapower's own results).  None can be produced in this container, so the test skips while the
    directory is empty."""
    files = _fixture_files()
    if not files:
        pytest.skip('no pandapower exports under fixtures/ (pandapower is not installed here)')
    for path in files:
        _check_export(path)


def _synthetic_export(net, path):
    """A file in the exporter's format whose "pandapower results" are the oracle's own — it pins nothing, it only
    lets the checker of real exports run here (matrix route, table route, every `out__` column)."""
    import sys, os
    from oracle import pd2ppc
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'scripts'))
    from export_pandapower_case import RESULTS, dump_tables
    sol = po.runpp(net, enforce_q_lims=True)
    ppc = sol['ppc']
    on = np.flatnonzero(sol['supplied'])                   # (pandapower's ppci holds the energised buses only)
    renum = {int(b): k for k, b in enumerate(on)}
    keep_br = [k for k in range(ppc.nbr) if ppc.status[k] and ppc.f[k] in renum and ppc.t[k] in renum]
    keep_g = [g for g in range(len(ppc.g_bus)) if ppc.g_status[g] and int(ppc.g_bus[g]) in renum]
    z0 = np.zeros(len(on))
    bus = np.column_stack([np.arange(len(on)), sol['bus_type'][on] if 'bus_type' in sol else ppc.bus_type[on], ppc.pd[on], ppc.qd[on],
                           ppc.gs[on], ppc.bs[on], z0 + 1, ppc.vm[on], ppc.va[on], ppc.base_kv[on], z0 + 1, z0 + 2, z0])
    bus[:, 1] = ppc.bus_type[on]
    branch = np.array([[renum[int(ppc.f[k])], renum[int(ppc.t[k])], ppc.r[k], ppc.x[k], ppc.b[k].real, 0, 0, 0, ppc.tap[k],
                        ppc.shift[k], 1, -360, 360] for k in keep_br], dtype=float)
    gen = np.array([[renum[int(ppc.g_bus[g])], ppc.g_p[g], 0.0, ppc.g_qmax[g], ppc.g_qmin[g], ppc.g_vg[g], 100, 1] for g in keep_g], dtype=float)
    v = sol['V'][on]
    data = dict(baseMVA=np.array(ppc.base_mva), bus=bus, branch=branch, gen=gen, res_vm=np.abs(v), res_va=np.degrees(np.angle(v)),
                br_g=np.array([-ppc.b[k].imag for k in keep_br]), enforce_q_lims=np.array(1))
    data.update(dump_tables(net))
    for tbl, cols in RESULTS:
        if tbl in net and len(net[tbl]):
            for col in cols:
                if col in net[tbl].columns:
                    data[f'out__{tbl}__{col}'] = net[tbl][col].to_numpy(dtype=float)
    np.savez_compressed(path, **data)


@pytest.mark.parametrize('code', ['mv-small', 'hv-small-sw', 'mv-3w'])
def test_export_checker_runs_on_a_synthetic_export(code, tmp_path):
    """The checker of real pandapower exports (`_check_export`) exercised end to end on a file in the exporter's
    format (tables with taps, switches, a three-winding transformer; pypower matrices; result tables) — so that the
    first real file a maintainer drops into fixtures/ meets tested code."""
    net, _ = grids.get_grid(code)
    path = tmp_path / f'{code}.npz'
    _synthetic_export(net, str(path))
    assert _check_export(str(path)) >= 3


def test_export_table_round_trip():
    """dump_tables / load_tables of scripts/export_pandapower_case.py (the part that needs no pandapower):
    a net survives the trip with an identical per-unit case."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'scripts'))
    from export_pandapower_case import dump_tables, load_tables
    from oracle import pd2ppc
    for code in ('hv-small-sw', 'mv-small'):
        net, _ = grids.get_grid(code)
        back = load_tables(dump_tables(net))
        a, b = pd2ppc.build_ppc(net), pd2ppc.build_ppc(back)
        assert a.nb == b.nb and a.nbr == b.nbr
        assert np.abs(po.make_ybus(a) - po.make_ybus(b)).max() == 0
        assert (po.make_sbus(a) == po.make_sbus(b)).all()
