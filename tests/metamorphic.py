"""Metamorphic equivalences of the pandapower element models (test code only).

VERDICT r05 #2: part of the converter and the solver is exercised by no number that pandapower published — bus-bus fusing,
taps on the lv side, `parallel`, the vector-group shift under `calculate_voltage_angles=True` (the mode every GPU config runs
in), shunts, several ext_grids.  Each of those has an EQUIVALENT formulation that goes through another, pinned code path:
two nets that pandapower's own model definitions make the same electrical problem must give the same answer.  No third-party
number is needed, and a wrong formula on one of the two paths shows up as a difference.

Every case is a function returning `(net_a, net_b, compare)`; `compare(res_a, res_b)` raises on a difference, where `res_x`
is the net after a power flow (`net.res_*` filled — by the oracle on the CPU, by the plug-in on the GPU).  The tests:
tests/test_metamorphic.py (oracle; plus stamp-level equality of the PRODUCT's converter where the two nets must compile to
the same admittances) and tests/test_gpu_metamorphic.py (the same pairs through `power_flow_solver(net)` = opfx_solve).
"""
import copy

import numpy as np

from opfgym_amd import grids, net as N

VM_TOL, VA_TOL, LD_TOL, S_TOL = 1e-9, 1e-7, 1e-6, 1e-6      # p.u., degree, percent, MW / Mvar


def _col(net, table, column):
    return net[table][column].to_numpy(float)


def _same(a, b, tol, what):
    a, b = np.asarray(a, float), np.asarray(b, float)
    assert a.shape == b.shape, what
    assert np.allclose(a, b, rtol=0, atol=tol, equal_nan=True), (what, float(np.nanmax(np.abs(a - b))))


def _angle_same(a_deg, b_deg, tol, what):
    d = np.deg2rad(np.asarray(a_deg, float) - np.asarray(b_deg, float))
    assert np.nanmax(np.abs(np.degrees(np.angle(np.exp(1j * d))))) < tol, what


def _compare_all(res_a, res_b, buses_a=None, buses_b=None, lines=True, trafos=True, ext=True):
    ba = res_a.res_bus if buses_a is None else res_a.res_bus.loc[buses_a]
    bb = res_b.res_bus if buses_b is None else res_b.res_bus.loc[buses_b]
    _same(ba.vm_pu, bb.vm_pu, VM_TOL, 'vm_pu')
    _angle_same(ba.va_degree, bb.va_degree, VA_TOL, 'va_degree')
    if lines:
        _same(_col(res_a, 'res_line', 'loading_percent'), _col(res_b, 'res_line', 'loading_percent'), LD_TOL, 'line loading')
    if trafos:
        _same(_col(res_a, 'res_trafo', 'loading_percent'), _col(res_b, 'res_trafo', 'loading_percent'), LD_TOL, 'trafo loading')
    if ext:
        _same(_col(res_a, 'res_ext_grid', 'p_mw'), _col(res_b, 'res_ext_grid', 'p_mw'), S_TOL, 'p_ext')
        _same(_col(res_a, 'res_ext_grid', 'q_mvar'), _col(res_b, 'res_ext_grid', 'q_mvar'), S_TOL, 'q_ext')


# ---------------------------------------------------------------------------------------------------------------------
def bus_bus_switch_is_one_bus():
    """A CLOSED bus-bus switch fuses its two buses: the same grid built with ONE bus in their place.  (mv-small: its best
    connected bus gets a twin behind a closed coupler; every second line end and unit moves to the twin.)"""
    base, _ = grids.get_grid('mv-small')
    a = copy.deepcopy(base)
    ends = list(a.line.from_bus) + list(a.line.to_bus)
    bar = int(max(set(ends), key=ends.count))                          # (the best connected bus: four line ends)
    twin = N.create_bus(a, float(a.bus.vn_kv.at[bar]))
    for c in a.bus.columns:
        if c != 'name':
            a.bus.at[twin, c] = a.bus.at[bar, c]
    moved = 0
    for tbl, col in (('line', 'from_bus'), ('line', 'to_bus'), ('load', 'bus'), ('sgen', 'bus')):
        for idx in a[tbl].index[a[tbl][col] == bar][::2]:
            a[tbl].at[idx, col] = twin
            moved += 1
    assert moved >= 2
    N.create_switch(a, bar, twin, 'b', closed=True)
    N.finalize(a)
    b = copy.deepcopy(base)                                            # (the grid as it was: `bar` is the one bus)

    def compare(ra, rb):
        _compare_all(ra, rb, buses_a=list(b.bus.index), buses_b=list(b.bus.index))
        assert abs(ra.res_bus.vm_pu.at[twin] - ra.res_bus.vm_pu.at[bar]) < 1e-15          # both halves report the fused bus
        assert abs(ra.res_bus.va_degree.at[twin] - ra.res_bus.va_degree.at[bar]) < 1e-12
    return a, b, compare


def _two_winding_pair(**trafo_kw):
    """110 kV slack - line - transformer - 20 kV load bus (- a second line and load), transformer parameters given."""
    net = N.Net()
    b0, b1, b2, b3 = N.create_bus(net, 110.), N.create_bus(net, 110.), N.create_bus(net, 20.), N.create_bus(net, 20.)
    N.create_ext_grid(net, b0, vm_pu=1.02, va_degree=0.0)
    N.create_line_from_parameters(net, b0, b1, 14.0, 0.06, 0.38, 9.5, 0.6)
    kw = dict(sn_mva=40.0, vn_hv_kv=110.0, vn_lv_kv=21.0, vk_percent=12.0, vkr_percent=0.35, pfe_kw=22.0, i0_percent=0.06,
              shift_degree=150.0)
    kw.update(trafo_kw)
    t = N.create_transformer_from_parameters(net, b1, b2, **kw)
    N.create_line_from_parameters(net, b2, b3, 3.2, 0.16, 0.12, 270.0, 0.36)
    N.create_load(net, b2, 9.0, 2.5)
    N.create_load(net, b3, 6.5, 1.8)
    N.create_sgen(net, b3, 2.0, -0.4)
    return N.finalize(net), t


def lv_side_tap_is_a_changed_lv_rating():
    """pandapower's tap changer scales the RATED voltage of its side (`_calc_tap_from_dataframe`: vn_lv * (1 + step% * (pos -
    neutral))) and everything downstream — ratio and the reference of the short-circuit impedance — uses the scaled rating:
    a transformer tapped on the lv side is the untapped transformer with vn_lv_kv changed accordingly."""
    a, t = _two_winding_pair(tap_side='lv', tap_neutral=0, tap_pos=-3, tap_step_percent=1.5)
    b, _ = _two_winding_pair(vn_lv_kv=21.0 * (1 + 0.015 * -3))
    # (the transformer's own loading_percent refers its current to the TABLE's rated voltage, which differs between the two
    #  nets by construction; the currents themselves are compared through the lines on both sides and the slack power)
    return a, b, lambda ra, rb: _compare_all(ra, rb, trafos=False)


def hv_side_tap_is_a_changed_hv_rating():
    a, t = _two_winding_pair(tap_side='hv', tap_neutral=0, tap_pos=4, tap_step_percent=1.25)
    b, _ = _two_winding_pair(vn_hv_kv=110.0 * (1 + 0.0125 * 4))
    return a, b, lambda ra, rb: _compare_all(ra, rb, trafos=False)


def parallel_two_is_two_elements():
    """`parallel = 2` on a line and on a transformer: two identical elements between the same buses; each carries half, so
    the loading of the pair equals the loading of either twin."""
    a, t = _two_winding_pair(parallel=2)
    a.line.at[a.line.index[1], 'parallel'] = 2
    b, _ = _two_winding_pair()
    tr = b.trafo.iloc[0]
    N.create_transformer_from_parameters(
        b, int(tr.hv_bus), int(tr.lv_bus), **{k: tr[k] for k in ('sn_mva', 'vn_hv_kv', 'vn_lv_kv', 'vk_percent', 'vkr_percent',
                                                                   'pfe_kw', 'i0_percent', 'shift_degree')})
    ln = b.line.iloc[1]
    N.create_line_from_parameters(b, int(ln.from_bus), int(ln.to_bus), ln.length_km, ln.r_ohm_per_km, ln.x_ohm_per_km,
                                  ln.c_nf_per_km, ln.max_i_ka)
    N.finalize(b)

    def compare(ra, rb):
        _compare_all(ra, rb, lines=False, trafos=False)
        _same(_col(ra, 'res_trafo', 'loading_percent')[0], _col(rb, 'res_trafo', 'loading_percent')[0], LD_TOL, 'trafo pair')
        _same(_col(rb, 'res_trafo', 'loading_percent')[0], _col(rb, 'res_trafo', 'loading_percent')[1], LD_TOL, 'trafo twins')
        _same(_col(ra, 'res_line', 'loading_percent')[:2], _col(rb, 'res_line', 'loading_percent')[:2], LD_TOL, 'line pair')
        _same(_col(rb, 'res_line', 'loading_percent')[1], _col(rb, 'res_line', 'loading_percent')[2], LD_TOL, 'line twins')
    return a, b, compare


def vector_group_shift_turns_the_angles_behind_it():
    """BASELINE configs 2 / 4 run the 144-bus MV stand-in with `calculate_voltage_angles=True` (110 kV slack) and 150 degree
    transformers.  A vector-group shift common to all transformers between the slack's level and a radial level below changes
    NO voltage magnitude, loading or slack power and turns every angle behind it by exactly the shift: the same grid with
    shift 0 and with shift 150."""
    base, _ = grids.get_grid('1-MV-urban--0-sw')
    a, b = copy.deepcopy(base), copy.deepcopy(base)
    assert (a.trafo.shift_degree == 150.0).all()
    b.trafo['shift_degree'] = 0.0
    hv = set(int(v) for v in a.trafo.hv_bus) | set(int(v) for v in a.ext_grid.bus)
    behind = [int(i) for i in a.bus.index if int(i) not in hv]

    def compare(ra, rb):
        _same(ra.res_bus.vm_pu, rb.res_bus.vm_pu, VM_TOL, 'vm_pu')
        _same(_col(ra, 'res_line', 'loading_percent'), _col(rb, 'res_line', 'loading_percent'), LD_TOL, 'line loading')
        _same(_col(ra, 'res_trafo', 'loading_percent'), _col(rb, 'res_trafo', 'loading_percent'), LD_TOL, 'trafo loading')
        _same(_col(ra, 'res_ext_grid', 'p_mw'), _col(rb, 'res_ext_grid', 'p_mw'), S_TOL, 'p_ext')
        _same(_col(ra, 'res_ext_grid', 'q_mvar'), _col(rb, 'res_ext_grid', 'q_mvar'), S_TOL, 'q_ext')
        _angle_same(ra.res_bus.va_degree.loc[behind], rb.res_bus.va_degree.loc[behind] - 150.0, VA_TOL, 'shifted angles')
        _angle_same(ra.res_bus.va_degree.loc[sorted(hv)], rb.res_bus.va_degree.loc[sorted(hv)], VA_TOL, 'hv angles')
    return a, b, compare


def shunt_is_a_constant_impedance_load():
    """A shunt (p_mw, q_mvar at vn_kv, `step` of them) draws P = p step |V|^2, Q = q step |V|^2 (|V| in p.u. of vn_kv): replaced
    by a constant-power load of exactly that size AT THE SOLVED VOLTAGE the power flow has the same solution.  Net b is built
    from net a's results (`compare` is handed both; the builder needs a solver — see `needs_first`)."""
    a, _ = _two_winding_pair()
    sh_bus = int(a.bus.index[3])
    N.create_shunt(a, sh_bus, q_mvar=-3.0, p_mw=0.4, step=2)
    N.finalize(a)

    def second(res_a):
        b, _ = _two_winding_pair()
        vm = float(res_a.res_bus.vm_pu.at[sh_bus])
        N.create_load(b, sh_bus, 0.4 * 2 * vm ** 2, -3.0 * 2 * vm ** 2)
        return N.finalize(b)
    return a, second, lambda ra, rb: _compare_all(ra, rb)


def two_ext_grids_at_one_set_point_are_a_fused_slack():
    """Two ext_grids with the same |V| and angle on two (non-adjacent) buses: both buses are held at the same complex voltage,
    so fusing them (a closed bus-bus switch, ONE ext_grid) changes nothing for the rest of the grid, and the two slack powers
    add up to the fused one's.  (hv-small keeps its own 380 kV ext_grid: three REF buses in net a, two in net b.)"""
    base, _ = grids.get_grid('hv-small')
    taken = set(int(v) for v in base.gen.bus) | set(int(v) for v in base.trafo.hv_bus) | set(int(v) for v in base.trafo.lv_bus)
    free = [int(i) for i in base.bus.index if int(i) not in taken and base.bus.vn_kv.at[i] == 110.0]
    x = free[3]
    near = set(int(v) for v in base.line.to_bus[base.line.from_bus == x]) | set(int(v) for v in base.line.from_bus[base.line.to_bus == x])
    y = next(i for i in free[8:] if i not in near and i != x)
    a = copy.deepcopy(base)
    N.create_ext_grid(a, x, vm_pu=1.01, va_degree=-2.0)
    N.create_ext_grid(a, y, vm_pu=1.01, va_degree=-2.0)
    N.finalize(a)
    b = copy.deepcopy(base)
    N.create_ext_grid(b, x, vm_pu=1.01, va_degree=-2.0)
    N.create_switch(b, x, y, 'b', closed=True)
    N.finalize(b)

    def compare(ra, rb):
        _same(ra.res_bus.vm_pu, rb.res_bus.vm_pu, VM_TOL, 'vm_pu')
        _angle_same(ra.res_bus.va_degree, rb.res_bus.va_degree, VA_TOL, 'va_degree')
        _same(_col(ra, 'res_line', 'loading_percent'), _col(rb, 'res_line', 'loading_percent'), LD_TOL, 'line loading')
        _same(_col(ra, 'res_trafo', 'loading_percent'), _col(rb, 'res_trafo', 'loading_percent'), LD_TOL, 'trafo loading')
        pa, pb = _col(ra, 'res_ext_grid', 'p_mw'), _col(rb, 'res_ext_grid', 'p_mw')
        qa, qb = _col(ra, 'res_ext_grid', 'q_mvar'), _col(rb, 'res_ext_grid', 'q_mvar')
        _same(pa[0], pb[0], S_TOL, 'slack P of the 380 kV ext_grid')
        _same(pa[1] + pa[2], pb[1], S_TOL, 'slack P of the pair')
        _same(qa[0], qb[0], S_TOL, 'slack Q of the 380 kV ext_grid')
        _same(qa[1] + qa[2], qb[1], S_TOL, 'slack Q of the pair')
        assert abs(pa[1]) > 1e-3 and abs(pa[2]) > 1e-3
    return a, b, compare


def open_bus_bus_switch_is_no_switch():
    """An OPEN bus-bus switch couples nothing: the grid without it."""
    base, _ = grids.get_grid('hv-small')
    a = copy.deepcopy(base)
    N.create_switch(a, int(a.bus.index[5]), int(a.bus.index[9]), 'b', closed=False)
    N.finalize(a)
    return a, copy.deepcopy(base), lambda ra, rb: _compare_all(ra, rb)


def ideal_phase_shifter_is_a_changed_vector_group():
    """An ideal phase shifter (`tap_phase_shifter`, `tap_step_degree` per step) only turns the angle: the same transformer with its
    vector-group shift changed by direction * (pos - neutral) * tap_step_degree (+ on the hv side, - on the lv side) and no tap.
    The two transformers in a LOOP with a plain line (an angle that is merely carried along a radial path would cancel out of
    every |V|): the circulating flow the shifter drives must be the same."""
    def grid(**kw):
        net, t = _two_winding_pair(shift_degree=0.0, **kw)
        # a second, plain transformer in parallel closes the loop the phase shifter acts on
        tr = dict(sn_mva=40.0, vn_hv_kv=110.0, vn_lv_kv=21.0, vk_percent=12.0, vkr_percent=0.35, pfe_kw=22.0, i0_percent=0.06, shift_degree=0.0)
        N.create_transformer_from_parameters(net, int(net.trafo.hv_bus.iloc[0]), int(net.trafo.lv_bus.iloc[0]), **tr)
        return N.finalize(net)
    out = []
    for side, sign in (('hv', 1.0), ('lv', -1.0)):
        a = grid(tap_side=side, tap_neutral=0, tap_pos=3, tap_step_degree=1.5, tap_phase_shifter=True)
        b = grid()
        b.trafo.at[b.trafo.index[0], 'shift_degree'] = sign * 3 * 1.5
        out.append((a, b))
    (a, b), (a2, b2) = out

    def compare(ra, rb):
        _compare_all(ra, rb)
        ld = _col(ra, 'res_trafo', 'loading_percent')
        assert abs(ld[0] - ld[1]) > 1.0                      # (the shifter does drive a circulating flow: the twins are loaded unequally)
    compare.second_pair = (a2, b2)
    return a, b, compare


def phase_shifter_given_in_percent_is_one_given_in_degrees():
    """An ideal phase shifter whose step is given as `tap_step_percent` (no `tap_step_degree`) turns the angle by 2 asin(du / 2),
    du = tap_step_percent / 100 x (pos - neutral) (pandapower `_calc_tap_from_dataframe`): the same shifter given in degrees with
    that angle per step — again in a loop with a plain transformer."""
    def grid(**kw):
        net, t = _two_winding_pair(shift_degree=0.0, **kw)
        tr = dict(sn_mva=40.0, vn_hv_kv=110.0, vn_lv_kv=21.0, vk_percent=12.0, vkr_percent=0.35, pfe_kw=22.0, i0_percent=0.06, shift_degree=0.0)
        N.create_transformer_from_parameters(net, int(net.trafo.hv_bus.iloc[0]), int(net.trafo.lv_bus.iloc[0]), **tr)
        return N.finalize(net)
    out = []
    for side, pos in (('hv', 3), ('lv', -2)):
        a = grid(tap_side=side, tap_neutral=0, tap_pos=pos, tap_step_percent=1.2, tap_phase_shifter=True)
        angle = np.degrees(2.0 * np.arcsin(abs(0.012 * pos) / 2.0)) * np.sign(pos)
        b = grid(tap_side=side, tap_neutral=0, tap_pos=pos, tap_step_percent=0.0, tap_step_degree=angle / pos, tap_phase_shifter=True)
        out.append((a, b))
    (a, b), (a2, b2) = out

    def compare(ra, rb):
        _compare_all(ra, rb)
        ld = _col(ra, 'res_trafo', 'loading_percent')
        assert abs(ld[0] - ld[1]) > 0.5                      # (a circulating flow: the shifter acts)
    compare.second_pair = (a2, b2)
    return a, b, compare


def storage_is_a_load():
    """A storage unit enters the power flow as a load of its p_mw, q_mvar (x scaling): positive = charging = consumption."""
    a, _ = _two_winding_pair()
    bus = int(a.bus.index[3])
    N.create_storage(a, bus, 1.7, 0.4, scaling=1.2)
    N.finalize(a)
    b, _ = _two_winding_pair()
    N.create_load(b, bus, 1.7 * 1.2, 0.4 * 1.2)
    N.finalize(b)
    return a, b, lambda ra, rb: _compare_all(ra, rb)


def line_conductance_is_a_shunt_at_each_end():
    """`g_us_per_km` of a line: the pi-model carries half of the total conductance G = g_us_per_km 1e-6 length parallel at each end —
    the same line without it plus, at both of its buses, a shunt of P = G / 2 x vn_kv^2 MW at rated voltage."""
    a, _ = _two_winding_pair()
    idx = a.line.index[1]
    a.line.at[idx, 'g_us_per_km'] = 85.0
    a.line.at[idx, 'parallel'] = 2
    N.finalize(a)
    b, _ = _two_winding_pair()
    b.line.at[idx, 'parallel'] = 2
    g_total = 85.0e-6 * float(b.line.length_km.at[idx]) * 2
    for bus in (int(b.line.from_bus.at[idx]), int(b.line.to_bus.at[idx])):
        N.create_shunt(b, bus, q_mvar=0.0, p_mw=g_total / 2 * float(b.bus.vn_kv.at[bus]) ** 2)
    N.finalize(b)

    def compare(ra, rb):
        # (the currents the line's own loading refers to include its conductance branches in `a` and not in `b`: compared through
        #  the other line, the transformer and the slack power)
        _compare_all(ra, rb, lines=False)
        _same(_col(ra, 'res_line', 'loading_percent')[0], _col(rb, 'res_line', 'loading_percent')[0], LD_TOL, 'the other line')
        assert abs(_col(ra, 'res_ext_grid', 'p_mw')[0] - _col(_two_winding_solved(), 'res_ext_grid', 'p_mw')[0]) > 1e-3      # (it has teeth: the losses are there)
    return a, b, compare


_SOLVED = {}


def _two_winding_solved():
    """The plain pair solved once by the oracle (a reference point for 'the element does something')."""
    if 'net' not in _SOLVED:
        from oracle import pf_oracle as po
        net, _ = _two_winding_pair()
        po.runpp(net, enforce_q_lims=False, calculate_voltage_angles=True)
        _SOLVED['net'] = net
    return _SOLVED['net']


def derating_factor_scales_the_loading():
    """`df` of a line multiplies its rated current, `df` of a transformer its rated power: the power flow itself is unchanged and
    `loading_percent` is the loading without the factor divided by it."""
    a, _ = _two_winding_pair(df=0.8)
    a.line.at[a.line.index[1], 'df'] = 0.6
    N.finalize(a)
    b, _ = _two_winding_pair()

    def compare(ra, rb):
        _compare_all(ra, rb, lines=False, trafos=False)
        _same(_col(ra, 'res_line', 'loading_percent'), _col(rb, 'res_line', 'loading_percent') / np.array([1.0, 0.6]), LD_TOL, 'line loading / df')
        _same(_col(ra, 'res_trafo', 'loading_percent'), _col(rb, 'res_trafo', 'loading_percent') / 0.8, LD_TOL, 'trafo loading / df')
    return a, b, compare


# ---- element types beyond the SimBench grids (round 6; VERDICT r05 "missing" #5) -----------------------------------------
def ward_is_a_load_and_a_shunt():
    """pandapower's ward: a constant-power part (ps_mw, qs_mvar) and a constant-impedance part (pz_mw, qz_mvar at 1 p.u. of the
    bus — a shunt whose rated voltage is the bus's own)."""
    a, _ = _two_winding_pair()
    bus = int(a.bus.index[3])
    N.create_ward(a, bus, ps_mw=1.1, qs_mvar=0.35, pz_mw=0.6, qz_mvar=-1.4)
    N.finalize(a)
    b, _ = _two_winding_pair()
    N.create_load(b, bus, 1.1, 0.35)
    N.create_shunt(b, bus, q_mvar=-1.4, p_mw=0.6)
    N.finalize(b)

    def compare(ra, rb):
        _compare_all(ra, rb)
        vm = float(ra.res_bus.vm_pu.at[bus])
        _same(ra.res_ward.p_mw, [1.1 + 0.6 * vm ** 2], S_TOL, 'res_ward.p_mw')
        _same(ra.res_ward.q_mvar, [0.35 - 1.4 * vm ** 2], S_TOL, 'res_ward.q_mvar')
        _same(ra.res_ward.vm_pu, [vm], VM_TOL, 'res_ward.vm_pu')
    return a, b, compare


def xward_is_a_ward_and_a_voltage_source_behind_an_impedance():
    """pandapower's extended ward: the ward's two parts at the bus plus, behind r_ohm + j x_ohm, an internal bus that a source
    without active power holds at vm_pu — built here from a ward, an extra bus, a line without capacitance and a generator
    with p_mw = 0."""
    a, _ = _two_winding_pair()
    bus = int(a.bus.index[3])
    N.create_xward(a, bus, ps_mw=0.9, qs_mvar=0.3, pz_mw=0.5, qz_mvar=-1.1, r_ohm=0.6, x_ohm=5.0, vm_pu=1.015)
    N.finalize(a)
    b, _ = _two_winding_pair()
    N.create_ward(b, bus, ps_mw=0.9, qs_mvar=0.3, pz_mw=0.5, qz_mvar=-1.1)
    inner = N.create_bus(b, 20.)
    N.create_line_from_parameters(b, bus, inner, 1.0, 0.6, 5.0, 0.0, 1.0)
    N.create_gen(b, inner, 0.0, vm_pu=1.015)
    N.finalize(b)

    def compare(ra, rb):
        common = list(ra.bus.index)
        _compare_all(ra, rb, buses_a=common, buses_b=common, lines=False)
        _same(_col(ra, 'res_line', 'loading_percent'), _col(rb, 'res_line', 'loading_percent')[:2], LD_TOL, 'line loading')
        vm = float(ra.res_bus.vm_pu.at[bus])
        i_ka = float(rb.res_line.loading_percent.iloc[2]) / 100.0 * 1.0          # (max_i_ka = 1)
        x = ra.res_xward.iloc[0]
        _same([x.vm_internal_pu], [1.015], VM_TOL, 'internal |V|')
        _same([x.vm_internal_pu, x.vm_pu], [float(rb.res_bus.vm_pu.at[inner]), vm], VM_TOL, 'voltages')
        _angle_same([x.va_internal_degree], [float(rb.res_bus.va_degree.at[inner])], VA_TOL, 'internal angle')
        # into the impedance: its losses (the source gives no active power) and, reactive, what the source does not deliver
        _same([x.p_mw], [0.9 + 0.5 * vm ** 2 + 3 * 0.6 * i_ka ** 2], S_TOL, 'res_xward.p_mw')
        _same([x.q_mvar], [0.3 - 1.1 * vm ** 2 + 3 * 5.0 * i_ka ** 2 - float(rb.res_gen.q_mvar.iloc[0])], S_TOL, 'res_xward.q_mvar')
    return a, b, compare


def dcline_is_two_generators():
    """pandapower runs a DC line as two generators (`_add_dcline_gens`): at the to bus one feeding in p_mw less the relative and
    the fixed losses at vm_to_pu, at the from bus one with -p_mw at vm_from_pu, each within its side's reactive range."""
    def base():
        net, _ = _two_winding_pair()
        far = N.create_bus(net, 20.)
        N.create_line_from_parameters(net, int(net.bus.index[3]), far, 5.0, 0.16, 0.12, 270.0, 0.36)
        N.create_load(net, far, 3.0, 0.8)
        return net, int(net.bus.index[2]), far
    a, near, far = base()
    N.create_dcline(a, near, far, p_mw=2.5, loss_percent=3.0, loss_mw=0.04, vm_from_pu=1.0, vm_to_pu=1.012,
                    min_q_from_mvar=-2.0, max_q_from_mvar=2.0, min_q_to_mvar=-0.2, max_q_to_mvar=0.2)
    N.create_dcline(a, far, near, p_mw=9.0, loss_percent=1.0, loss_mw=0.0, vm_from_pu=1.0, vm_to_pu=1.0, in_service=False)
    N.finalize(a)
    b, near, far = base()
    N.create_gen(b, far, 2.5 * 0.97 - 0.04, vm_pu=1.012, min_q_mvar=-0.2, max_q_mvar=0.2)
    N.create_gen(b, near, -2.5, vm_pu=1.0, min_q_mvar=-2.0, max_q_mvar=2.0)
    N.finalize(b)

    def compare(ra, rb):
        _compare_all(ra, rb)
        d, g = ra.res_dcline, rb.res_gen
        _same(d.p_from_mw, [2.5, 0.0], S_TOL, 'p_from_mw')
        _same(d.p_to_mw, [-(2.5 * 0.97 - 0.04), 0.0], S_TOL, 'p_to_mw')
        _same(d.pl_mw, [2.5 * 0.03 + 0.04, 0.0], S_TOL, 'pl_mw')
        _same(d.q_to_mvar, [-float(g.q_mvar.iloc[0]), 0.0], S_TOL, 'q_to_mvar')
        _same(d.q_from_mvar, [-float(g.q_mvar.iloc[1]), 0.0], S_TOL, 'q_from_mvar')
        _same([d.vm_to_pu.iloc[0], d.vm_from_pu.iloc[0]], [float(ra.res_bus.vm_pu.at[far]), float(ra.res_bus.vm_pu.at[near])], VM_TOL, 'end voltages')
    return a, b, compare


def motor_is_a_load():
    """pandapower's motor: P = pn_mech / efficiency x loading x scaling, Q from cos_phi (inductive)."""
    a, _ = _two_winding_pair()
    bus = int(a.bus.index[2])
    N.create_motor(a, bus, pn_mech_mw=2.4, cos_phi=0.86, efficiency_percent=93.0, loading_percent=70.0, scaling=1.5)
    N.create_motor(a, bus, pn_mech_mw=9.0, cos_phi=0.8, in_service=False)
    N.finalize(a)
    p = 2.4 / 0.93 * 0.70 * 1.5
    q = p * np.tan(np.arccos(0.86))
    b, _ = _two_winding_pair()
    N.create_load(b, bus, p, q)
    N.finalize(b)

    def compare(ra, rb):
        _compare_all(ra, rb)
        _same(ra.res_motor.p_mw, [p, 0.0], S_TOL, 'res_motor.p_mw')
        _same(ra.res_motor.q_mvar, [q, 0.0], S_TOL, 'res_motor.q_mvar')
    return a, b, compare


def _line_gap_pair():
    """The 110 / 20 kV pair with the 20 kV line cut: its far half hangs on a new bus `gap` that the case connects to bus 2."""
    net, _ = _two_winding_pair()
    gap = N.create_bus(net, 20.)
    net.line.at[net.line.index[1], 'from_bus'] = gap
    return net, int(net.bus.index[2]), gap


def symmetric_impedance_is_a_line_without_charging():
    """A series impedance z (p.u. of its own sn_mva) with the same value in both directions is a line of z sn_net / sn_imp
    x vn^2 / sn_net ohm without capacitance."""
    a, near, gap = _line_gap_pair()
    N.create_impedance(a, near, gap, rft_pu=0.004, xft_pu=0.011, sn_mva=25.0)
    N.finalize(a)
    b, near, gap = _line_gap_pair()
    z_base = 20.0 ** 2 / 25.0
    N.create_line_from_parameters(b, near, gap, 1.0, 0.004 * z_base, 0.011 * z_base, 0.0, 1.0)
    N.finalize(b)

    def compare(ra, rb):
        _compare_all(ra, rb, lines=False)
        _same(_col(ra, 'res_line', 'loading_percent'), _col(rb, 'res_line', 'loading_percent')[:2], LD_TOL, 'line loading')
        # (the flows of the impedance: what leaves the near bus arrives, less the losses r |I|^2, at the other)
        r = ra.res_impedance
        i_ka = float(r.i_from_ka.iloc[0])
        assert abs(i_ka - float(r.i_to_ka.iloc[0])) < 1e-9
        _same(r.pl_mw, [3 * (0.004 * z_base) * i_ka ** 2], S_TOL, 'pl_mw = 3 r I^2')
        _same(r.ql_mvar, [3 * (0.011 * z_base) * i_ka ** 2], S_TOL, 'ql_mvar = 3 x I^2')
        _same(_col(rb, 'res_line', 'loading_percent')[2], i_ka / 1.0 * 100.0, LD_TOL, 'i_ka of the twin line')
    return a, b, compare


def bus_bus_switch_with_impedance_is_a_short_line():
    """A CLOSED bus-bus switch with z_ohm > 0 does not fuse its buses: pandapower inserts a branch of |z| = z_ohm whose r / x
    ratio is `runpp`'s switch_rx_ratio (default 2) — r = 2 z / sqrt(5), x = z / sqrt(5)."""
    a, near, gap = _line_gap_pair()
    N.create_switch(a, near, gap, 'b', closed=True, z_ohm=0.08)
    N.finalize(a)
    b, near, gap = _line_gap_pair()
    N.create_line_from_parameters(b, near, gap, 1.0, 0.08 * 2 / np.sqrt(5.0), 0.08 / np.sqrt(5.0), 0.0, 1.0)
    N.finalize(b)

    def compare(ra, rb):
        _compare_all(ra, rb, lines=False)
        _same(_col(ra, 'res_line', 'loading_percent'), _col(rb, 'res_line', 'loading_percent')[:2], LD_TOL, 'line loading')
    return a, b, compare


CASES = {f.__name__: f for f in (phase_shifter_given_in_percent_is_one_given_in_degrees, line_conductance_is_a_shunt_at_each_end, derating_factor_scales_the_loading, ward_is_a_load_and_a_shunt, xward_is_a_ward_and_a_voltage_source_behind_an_impedance, dcline_is_two_generators, motor_is_a_load, symmetric_impedance_is_a_line_without_charging,
                                 bus_bus_switch_with_impedance_is_a_short_line,
                                 ideal_phase_shifter_is_a_changed_vector_group, storage_is_a_load, bus_bus_switch_is_one_bus, lv_side_tap_is_a_changed_lv_rating, hv_side_tap_is_a_changed_hv_rating,
                                 parallel_two_is_two_elements, vector_group_shift_turns_the_angles_behind_it,
                                 shunt_is_a_constant_impedance_load, two_ext_grids_at_one_set_point_are_a_fused_slack,
                                 open_bus_bus_switch_is_no_switch)}
# pairs that must compile to the SAME bus admittance matrix in the product's converter (no solve needed to compare them)
SAME_ADMITTANCES = ('phase_shifter_given_in_percent_is_one_given_in_degrees', 'line_conductance_is_a_shunt_at_each_end', 'derating_factor_scales_the_loading', 'ward_is_a_load_and_a_shunt', 'dcline_is_two_generators', 'motor_is_a_load', 'symmetric_impedance_is_a_line_without_charging',
                    'bus_bus_switch_with_impedance_is_a_short_line', 'ideal_phase_shifter_is_a_changed_vector_group', 'storage_is_a_load', 'lv_side_tap_is_a_changed_lv_rating', 'hv_side_tap_is_a_changed_hv_rating', 'parallel_two_is_two_elements',
                    'open_bus_bus_switch_is_no_switch', 'bus_bus_switch_is_one_bus')


def run(case, solve):
    """Solve both nets of a case with `solve(net)` (in place: fills net.res_*) and compare."""
    a, b, compare = CASES[case]()
    solve(a)
    if callable(b):
        b = b(a)
    solve(b)
    compare(a, b)
    if hasattr(compare, 'second_pair'):          # (a case may carry a second pair checked the same way)
        a2, b2 = compare.second_pair
        solve(a2)
        solve(b2)
        compare(a2, b2)
