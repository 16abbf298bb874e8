"""Metamorphic equivalences of the element models on the CPU (tests/metamorphic.py): the oracle solves both
nets of every pair; where the two nets must compile to the same admittances the PRODUCT's converter (`case.net_to_case`) is
compared at the stamp level — no solver needed.  The same pairs run through the GPU in tests/test_gpu_metamorphic.py."""
import numpy as np
import pytest

import metamorphic
from opfgym_amd.case import net_to_case
from oracle import pf_oracle as po


@pytest.mark.parametrize('case', list(metamorphic.CASES))
def test_equivalent_formulations_give_the_same_power_flow_on_the_oracle(case):
    metamorphic.run(case, lambda net: po.runpp(net, enforce_q_lims=False, calculate_voltage_angles=True))


@pytest.mark.parametrize('case', metamorphic.SAME_ADMITTANCES)
def test_equivalent_formulations_compile_to_the_same_admittances(case):
    """Product converter: the bus admittance matrices of the two nets are equal (bus order: the net's; the fused pair of the
    bus-bus case compares the common buses)."""
    a, b, _ = metamorphic.CASES[case]()
    ca, cb = net_to_case(a), net_to_case(b)
    ya, yb = ca.ybus_dense(), cb.ybus_dense()
    common = [i for i in b.bus.index if int(i) in ca.bus_lookup and int(i) in cb.bus_lookup]
    ia = [ca.bus_lookup[int(i)] for i in common]
    ib = [cb.bus_lookup[int(i)] for i in common]
    assert len(set(ia)) == len(ia) == cb.nb
    assert np.allclose(ya[np.ix_(ia, ia)], yb[np.ix_(ib, ib)], rtol=0, atol=1e-9 * max(1.0, np.abs(yb).max()))
    assert ca.nb == cb.nb


def test_a_wrong_tap_model_would_be_noticed():
    """The equivalences have teeth: a transformer whose lv tap is applied to the ratio but NOT to the impedance reference (a
    plausible mis-model) differs from the changed-rating net by far more than the tolerances."""
    a, b, _ = metamorphic.CASES['lv_side_tap_is_a_changed_lv_rating']()
    po.runpp(a, enforce_q_lims=False)
    wrong = metamorphic._two_winding_pair(vn_lv_kv=21.0 * (1 + 0.015 * -3), vk_percent=12.0 * (1 + 0.015 * -3) ** 2)[0]
    po.runpp(wrong, enforce_q_lims=False)
    assert np.abs(a.res_bus.vm_pu.to_numpy() - wrong.res_bus.vm_pu.to_numpy()).max() > 1e-5
