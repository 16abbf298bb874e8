"""CPU: pypower-format case import (the route for real pandapower/SimBench exports,
SURVEY §8f N3) gives the same per-unit case — and the same power flow — as the
table route for the WSCC 9-bus system typed in MATPOWER `case9` layout."""
import numpy as np

from helpers import oracle_ppc_solve
from opfgym_amd import grids
from opfgym_amd.case import net_to_case
from opfgym_amd.ppci_io import case_from_ppc
from oracle import pf_oracle as po


def _ppc_case9():
    bus = np.zeros((9, 13))
    bus[:, 0] = np.arange(9)
    bus[:, 1] = [3, 2, 2, 1, 1, 1, 1, 1, 1]
    bus[[4, 6, 8], 2] = [90, 100, 125]
    bus[[4, 6, 8], 3] = [30, 35, 50]
    bus[:, 7], bus[:, 9] = 1.0, 345.0
    branch = np.zeros((9, 13))
    for k, (f, t, r, x, b) in enumerate(grids._CASE9_BRANCH):
        branch[k, :5] = [f - 1, t - 1, r, x, b]
        branch[k, 5], branch[k, 10] = 250.0, 1
    gen = np.zeros((3, 10))
    gen[:, 0] = [0, 1, 2]
    gen[:, 1] = [0, 163, 85]
    gen[:, 3], gen[:, 4] = 300, -300
    gen[:, 5] = [1.04, 1.025, 1.025]
    gen[:, 7] = 1
    return 100.0, bus, branch, gen


def test_ppc_route_equals_table_route():
    """Product: matrices -> Case (ppci_io) vs tables -> Case (net_to_case): same admittances.  Oracle: its own
    matrix reader vs its own table converter: same solution, and the published voltage profile."""
    case, p, q, qmin, qmax = case_from_ppc(*_ppc_case9())
    net = grids.case9()
    ref_case = net_to_case(net)
    assert np.abs(case.ybus_dense() - ref_case.ybus_dense()).max() < 1e-9
    sol = oracle_ppc_solve(*_ppc_case9())
    ref = po.runpp(net, enforce_q_lims=False)
    assert sol['converged']
    assert np.abs(sol['V'] - ref['V']).max() < 1e-10
    assert np.abs(np.abs(sol['V']) - grids.CASE9_VM).max() < 1e-3
    # the product's loading scale of a matrix case: percent of RATE_A at nominal voltage
    assert np.allclose(case.kf, 100.0 / 250.0 * 100.0) and np.allclose(case.kt, case.kf)
