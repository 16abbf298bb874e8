"""CPU: pandapower element types beyond those of the SimBench grids — ward, motor, impedance, closed bus-bus switch with
z_ohm — in both converters (product `case.net_to_case`, oracle `pd2ppc.build_ppc`).  The equivalences to elements that are
pinned otherwise live in tests/metamorphic.py; here: the one thing without an equivalent element (different impedances per
direction) by hand, the refusals that stay, and the static injections of the batched environment."""
import copy

import numpy as np
import pytest

import beyond_simbench
from opfgym_amd import net as N
from opfgym_amd.case import KIND_IMPEDANCE, KIND_SWITCH, bus_injections, net_to_case, static_consumption
from oracle import pd2ppc
from oracle import pf_oracle as po


def _two_bus(**imp):
    net = N.Net(sn_mva=10.0)
    b0, b1 = N.create_bus(net, 20.0), N.create_bus(net, 20.0)
    N.create_ext_grid(net, b0, 1.01)
    N.create_impedance(net, b0, b1, **imp)
    N.create_load(net, b1, 2.0, 0.5)
    return N.finalize(net)


def test_impedance_with_different_values_per_direction_by_hand():
    """pandapower's makeYbus for BR_R_ASYM / BR_X_ASYM: the from side sees z_ft, the to side z_tf, both referred from the
    element's sn_mva to the net's: Yff = -Yft = 1 / z_ft, Ytt = -Ytf = 1 / z_tf."""
    net = _two_bus(rft_pu=0.01, xft_pu=0.04, sn_mva=5.0, rtf_pu=0.02, xtf_pu=0.05)
    z_ft, z_tf = (0.01 + 0.04j) * 10.0 / 5.0, (0.02 + 0.05j) * 10.0 / 5.0
    want = np.array([[1 / z_ft, -1 / z_ft], [-1 / z_tf, 1 / z_tf]])
    case = net_to_case(net)
    assert (case.br_kind == [KIND_IMPEDANCE]).all() and case.kf[0] == 0.0 == case.kt[0]
    assert np.allclose(case.ybus_dense(), want, rtol=0, atol=1e-12)
    assert np.allclose(po.make_ybus(pd2ppc.build_ppc(net)).toarray(), want, rtol=0, atol=1e-12)
    # the solved flows satisfy each side's own equation, and the result table reports them
    po.runpp(net)
    v = net.res_bus.vm_pu.to_numpy() * np.exp(1j * np.deg2rad(net.res_bus.va_degree.to_numpy()))
    s_from = v[0] * np.conj((v[0] - v[1]) / z_ft) * 10.0
    s_to = v[1] * np.conj((v[1] - v[0]) / z_tf) * 10.0
    r = net.res_impedance.iloc[0]
    assert np.allclose([r.p_from_mw, r.q_from_mvar, r.p_to_mw, r.q_to_mvar], [s_from.real, s_from.imag, s_to.real, s_to.imag], atol=1e-9)
    assert abs(s_to.real + 2.0) < 1e-6 and abs(s_to.imag + 0.5) < 1e-6          # (what the load takes)
    assert abs(r.i_from_ka - abs(s_from) / (np.sqrt(3) * abs(v[0]) * 20.0)) < 1e-12


def test_switch_impedance_uses_pandapowers_rx_ratio():
    net = N.Net(sn_mva=1.0)
    b0, b1 = N.create_bus(net, 10.0), N.create_bus(net, 10.0)
    N.create_ext_grid(net, b0)
    N.create_switch(net, b0, b1, 'b', closed=True, z_ohm=0.5)
    N.create_load(net, b1, 0.1)
    N.finalize(net)
    case = net_to_case(net)
    assert case.nb == 2 and (case.br_kind == [KIND_SWITCH]).all()
    z = (0.5 * 2 / np.sqrt(5) + 1j * 0.5 / np.sqrt(5)) / (10.0 ** 2 / 1.0)
    assert abs(abs(z) - 0.5 / 100.0) < 1e-15
    assert np.allclose(case.ybus_dense(), np.array([[1, -1], [-1, 1]]) / z, rtol=1e-12)
    # z_ohm = 0 and an OPEN switch with z_ohm: as before — one bus, resp. no connection (the load bus is de-energised)
    net.switch['z_ohm'] = 0.0
    assert net_to_case(net).nb == 1
    net.switch['z_ohm'], net.switch['closed'] = 0.5, False
    assert net_to_case(net).nb == 1 and b1 not in net_to_case(net).bus_lookup


def test_static_consumption_and_injections_of_the_helper_grid():
    net, _ = beyond_simbench.grid()
    static = static_consumption(net)
    assert np.allclose(static['ward'][0], [0.12, 0.0, 0.0]) and np.allclose(static['ward'][1], [0.04, 0.0, 0.0])
    p_m = 0.25 / 0.94 * 0.8 * 1.2
    assert np.allclose(static['motor'][0], [p_m, 0.0]) and np.allclose(static['motor'][1], [p_m * np.tan(np.arccos(0.87)), 0.0])
    case = net_to_case(net)
    assert np.allclose(static['xward'][0], [0.06, 0.0]) and np.allclose(static['xward'][1], [0.02, 0.0])
    plain = copy.deepcopy(net)
    for tbl in ('ward', 'motor', 'xward'):
        plain[tbl] = plain[tbl].iloc[0:0]
    p1, q1 = bus_injections(net, case)[:2]
    p0, q0 = bus_injections(plain, net_to_case(plain))[:2]
    assert abs(p0.sum() - p1.sum() - (0.12 + 0.06 + p_m)) < 1e-12 and abs(q0.sum() - q1.sum() - (0.04 + 0.02 + static['motor'][1][0])) < 1e-12
    # the wards' constant-impedance parts are shunts at 1 p.u. of their bus
    assert abs(case.gs.sum() * case.base_mva - (0.08 + 0.04)) < 1e-12 and abs(case.bs.sum() * case.base_mva - (0.15 + 0.3 + 0.05)) < 1e-12
    # the in-service xward's internal source: one auxiliary PV bus at its set-point behind the impedance
    (pos, aux), = case.meta['xward_bus'].items()
    assert pos == 0 and case.bus_type[aux] == 2 and case.vm_set[aux] == 1.01 and aux not in case.bus_lookup.values()


@pytest.mark.parametrize('label,mutate', [
    ('svc', lambda net: net.__setitem__('svc', net['ward'].assign(x_l_ohm=1.0, x_cvar_ohm=-10.0, set_vm_pu=1.0))),
    ('z_ohm', lambda net: (N.create_switch(net, int(net.line.from_bus.iloc[0]), int(net.line.index[0]), 'l', closed=True, z_ohm=0.1), N.finalize(net))),
    ('slack', lambda net: net.gen.__setitem__('slack', True) if len(net.gen) else net.__setitem__('gen', net['gen'])),
])
def test_what_stays_refused(label, mutate):
    net, _ = beyond_simbench.grid()
    if label == 'slack':
        N.create_gen(net, int(net.bus.index[5]), 0.5)
        N.finalize(net)
    mutate(net)
    with pytest.raises(ValueError, match=label):
        net_to_case(net)
    with pytest.raises(ValueError, match=label):
        pd2ppc.build_ppc(net)


@pytest.mark.parametrize('enforce', [False, True])
def test_the_table_writer_reports_the_new_elements_as_the_oracle_does(enforce):
    """The plug-in's table writer (`solver_plugin._write_results`) fed with per-BUS results taken from the oracle's solution
    writes the oracle's own `res_ward / res_xward / res_motor / res_impedance / res_dcline` — and `res_gen` where a DC line ends
    on the bus of a generator (its end joins pfsoln's split of the bus's reactive power) — on the CPU, no solver involved
    (the same helper as tests/test_generator_dispatch.py)."""
    from test_generator_dispatch import tables_from_per_bus_results
    from opfgym_amd import grids
    net, _ = grids.get_grid('hv-small')
    beyond_simbench.add_elements(net)
    gen_bus = int(net.gen.bus.iloc[1])
    net.gen.loc[net.gen.index[1], ['min_q_mvar', 'max_q_mvar']] = [-30.0, 40.0]
    other = int(net.bus.index[net.bus.vn_kv == net.bus.vn_kv.at[gen_bus]][5])
    N.create_dcline(net, other, gen_bus, p_mw=6.0, loss_percent=2.0, loss_mw=0.1, vm_from_pu=1.0, vm_to_pu=float(net.gen.vm_pu.iloc[1]),
                    min_q_from_mvar=-5.0, max_q_from_mvar=5.0, min_q_to_mvar=-10.0, max_q_to_mvar=20.0)
    N.finalize(net)
    out, ref, sol = tables_from_per_bus_results(net, enforce)
    checked = 0
    for tbl in ('res_gen', 'res_ext_grid', 'res_ward', 'res_xward', 'res_motor', 'res_impedance', 'res_dcline'):
        for col in ref[tbl].columns:
            a, b = out[tbl][col].to_numpy(float), ref[tbl][col].to_numpy(float)
            assert a.shape == b.shape and np.allclose(a, b, rtol=0, atol=1e-6, equal_nan=True), (tbl, col, a, b)
            checked += 1
    assert checked >= 25
    # the line's to end and the generator share the bus's reactive power by their ranges: neither reports the bus total
    q_gen, q_to = float(ref.res_gen.q_mvar.iloc[1]), -float(ref.res_dcline.q_to_mvar.iloc[-1])
    assert abs(q_gen) > 1e-3 and abs(q_to) > 1e-3 and abs(q_gen - q_to) > 1e-3
