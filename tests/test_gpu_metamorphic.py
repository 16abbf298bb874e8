"""GPU: the metamorphic equivalences of tests/metamorphic.py through `power_flow_solver(net)` (= net_to_case + opfx_solve),
and each GPU result against the oracle's on the same net."""
import copy

import numpy as np
import pytest

import metamorphic

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('case', list(metamorphic.CASES))
def test_equivalent_formulations_give_the_same_power_flow_on_the_gpu(case):
    from opfgym_amd import power_flow_solver
    metamorphic.run(case, lambda net: power_flow_solver(net, enforce_q_lims=False, calculate_voltage_angles=True))


@pytest.mark.parametrize('case', list(metamorphic.CASES))
def test_each_formulation_matches_the_oracle(case):
    from opfgym_amd import power_flow_solver
    from oracle import pf_oracle as po
    a, b, _ = metamorphic.CASES[case]()
    nets = [a]
    if not callable(b):
        nets.append(b)
    for net in nets:
        ref = copy.deepcopy(net)
        po.runpp(ref, enforce_q_lims=False, calculate_voltage_angles=True)
        power_flow_solver(net, enforce_q_lims=False, calculate_voltage_angles=True)
        assert np.allclose(net.res_bus.vm_pu, ref.res_bus.vm_pu, rtol=0, atol=1e-8, equal_nan=True)
        d = np.deg2rad(net.res_bus.va_degree.to_numpy(float) - ref.res_bus.va_degree.to_numpy(float))
        assert np.nanmax(np.abs(np.angle(np.exp(1j * d)))) < 1e-8
        for tbl in ('res_line', 'res_trafo'):
            assert np.allclose(net[tbl].loading_percent, ref[tbl].loading_percent, rtol=0, atol=1e-6, equal_nan=True)
        assert np.allclose(net.res_ext_grid.p_mw, ref.res_ext_grid.p_mw, rtol=0, atol=1e-6)
        assert np.allclose(net.res_ext_grid.q_mvar, ref.res_ext_grid.q_mvar, rtol=0, atol=1e-6)
