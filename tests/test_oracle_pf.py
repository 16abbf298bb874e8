"""CPU: pin the power-flow oracle (oracle/pf_oracle.py) with known answers.
pandapower is unavailable, and the reference's tests hold no power-flow result
(SURVEY §8c), so the pins are: the closed-form two-bus solution, the published
WSCC 9-bus load flow, and algebraic self-checks."""
import numpy as np
import pytest

from opfgym_amd import grids
from opfgym_amd.case import net_to_case
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
    case = net_to_case(net)
    sol = po.runpp(net, enforce_q_lims=False)
    v = sol['V']
    p, q, *_ = po.bus_injections(net, case)
    mis = v * np.conj(sol['ybus'] @ v) - (p + 1j * q) / case.base_mva
    free = case.bus_type != 3
    assert np.abs(mis.real[free]).max() < 1e-8
    assert np.abs(mis.imag[case.bus_type == 1]).max() < 1e-8
    # power balance: slack + injections = branch losses
    br = po.branch_results(case, v)
    losses = (br['s_from'] + br['s_to']).sum() * case.base_mva
    s_bus = (v * np.conj(sol['ybus'] @ v)).sum() * case.base_mva
    shunt = (np.abs(v) ** 2 * (case.gs - 1j * case.bs)).sum() * case.base_mva
    assert abs(s_bus - losses - shunt) < 1e-6
    assert losses.real > 0


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
    from helpers import ieee14_ppc
    from opfgym_amd.ppci_io import case_from_ppc
    base, bus, branch, gen, pub = ieee14_ppc()
    case, p, q, _, _ = case_from_ppc(base, bus, branch, gen)
    sol = po.solve_case(case, p, q)
    v = sol['V']
    assert sol['converged'] and sol['iterations'] <= 5
    assert np.abs(np.abs(v) - pub['vm']).max() < 6e-4            # published to 3 decimals
    assert np.abs(np.degrees(np.angle(v)) - pub['va_deg']).max() < 1e-3
    s = v * np.conj(sol['ybus'] @ v) * base
    assert abs(s[0].real + bus[0, 2] - pub['p_slack_mw']) < 0.01
    assert abs(s[0].imag + bus[0, 3] - pub['q_slack_mvar']) < 0.01
    assert abs(s.real.sum() - pub['losses_mw']) < 0.01
