"""Test helper: a grid with the pandapower element types that no SimBench grid holds and round 6 added to both converters —
wards, extended wards, motors, DC lines, series impedances (one with different values per direction), a closed bus-bus switch with an impedance."""
import numpy as np

from opfgym_amd import grids, net as N


def z_scale_of(kv):
    return kv ** 2 / 400.0              # (ohm / per-unit values below are meant for 20 kV)


def add_elements(net):
    """In place on a grid with one big voltage level (the `mv-small` stand-in: 32 buses at 20 kV; `hv-small`, LV grids alike:
    buses are picked by position within that level): returns the net."""
    kv = float(net.bus.vn_kv.value_counts().index[0])
    lvl = [int(b) for b in net.bus.index[net.bus.vn_kv == kv]]
    at = lambda k: lvl[k % len(lvl)]
    N.create_ward(net, at(5), ps_mw=0.12, qs_mvar=0.04, pz_mw=0.08, qz_mvar=-0.15)
    N.create_ward(net, at(11), ps_mw=0.05, qs_mvar=-0.01, pz_mw=0.02, qz_mvar=0.03, in_service=False)
    N.create_ward(net, at(17), ps_mw=0.0, qs_mvar=0.0, pz_mw=0.0, qz_mvar=-0.3)            # (a pure capacitor bank)
    N.create_motor(net, at(8), pn_mech_mw=0.25, cos_phi=0.87, efficiency_percent=94.0, loading_percent=80.0, scaling=1.2)
    N.create_motor(net, at(20), pn_mech_mw=0.4, cos_phi=0.8, in_service=False)
    # an extended ward: its internal source holds a bus behind an impedance at 1.01 p.u. (an auxiliary PV bus of the case)
    N.create_xward(net, at(22), ps_mw=0.06, qs_mvar=0.02, pz_mw=0.04, qz_mvar=-0.05, r_ohm=0.8 * z_scale_of(kv), x_ohm=6.0 * z_scale_of(kv),
                   vm_pu=1.01)
    N.create_xward(net, at(13), ps_mw=0.1, qs_mvar=0.0, pz_mw=0.0, qz_mvar=0.0, r_ohm=1.0, x_ohm=5.0, vm_pu=1.0, in_service=False)
    # a DC line between two feeders: two generators in the power flow (to bus first), each holding its bus voltage within a range
    N.create_dcline(net, at(16), at(27), p_mw=0.2, loss_percent=2.5, loss_mw=0.005, vm_from_pu=1.02, vm_to_pu=1.018,
                    min_q_from_mvar=-0.15, max_q_from_mvar=0.15, min_q_to_mvar=-0.1, max_q_to_mvar=0.1)
    N.create_dcline(net, at(2), at(19), p_mw=0.5, loss_percent=1.0, loss_mw=0.0, vm_from_pu=1.0, vm_to_pu=1.0, in_service=False)
    # a tie between two feeders as a series impedance whose two directions differ, a symmetric one in parallel to a line
    z_scale = kv ** 2 / 400.0                          # (the per-unit values below are meant for 20 kV on the element's sn_mva)
    N.create_impedance(net, at(7), at(14), rft_pu=0.012, xft_pu=0.03, sn_mva=10.0 * z_scale, rtf_pu=0.015, xtf_pu=0.036)
    ln = net.line[(net.line.from_bus.isin(lvl)) & (net.line.to_bus.isin(lvl))].iloc[3]
    N.create_impedance(net, int(ln.from_bus), int(ln.to_bus), rft_pu=0.02, xft_pu=0.05, sn_mva=5.0 * z_scale)
    N.create_impedance(net, at(3), at(25), rft_pu=0.01, xft_pu=0.02, sn_mva=10.0 * z_scale, in_service=False)
    # a consumer behind a closed bus-bus switch with a contact resistance
    far = N.create_bus(net, kv)
    for c in net.bus.columns:
        if c not in ('name', 'vn_kv'):
            net.bus.at[far, c] = net.bus.at[at(9), c]
    moved = net.load.index[net.load.bus == at(9)]
    moved = moved[0] if len(moved) else net.load.index[4]          # (an existing load: the profiles know its index)
    net.load.at[moved, 'bus'] = far
    N.create_switch(net, at(9), far, 'b', closed=True, z_ohm=0.35 * z_scale)
    return N.finalize(net)


def grid():
    net, prof = grids.get_grid('mv-small')
    return add_elements(net), prof
