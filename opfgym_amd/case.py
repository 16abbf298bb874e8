"""Per-unit power-flow case and the net→case conversion.

A :class:`Case` is what crosses the C ABI (`include/opfx.h: opfx_case`): a
MATPOWER/pypower-shaped per-unit description of one grid topology — buses with
types and set-points, branches as their four admittance-matrix stamps, and the
scale factors that turn per-unit branch currents into `loading_percent`.

`net_to_case` restates the third-party pandapower `_pd2ppc` conversion
 for the element types that occur in the SimBench
benchmark grids: buses, lines, two-winding transformers, loads, sgens,
storages, gens, ext_grids, shunts, bus-bus/line/trafo switches, three-winding
transformers (star equivalent with an auxiliary bus) — and, beyond those
grids (round 6), wards, extended wards, series impedances, motors, DC lines and
closed bus-bus switches with an impedance.  pandapower is
not importable here, so the element formulas are restated from its published
documentation ("Electric model" pages of line / trafo); parity of this
conversion against pandapower itself is NOT verified in this repository (see
DESIGN.md "parity unpinned").
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

PQ, PV, REF = 1, 2, 3
KIND_LINE, KIND_TRAFO, KIND_TRAFO3W, KIND_IMPEDANCE, KIND_SWITCH, KIND_XWARD = 0, 1, 2, 3, 4, 5
# pandapower `runpp(switch_rx_ratio=2)`: a closed bus-bus switch with z_ohm > 0 is a branch of |z| = z_ohm with r / x = 2
SWITCH_RX_RATIO = 2.0


@dataclass
class Case:
    base_mva: float
    # buses (internal consecutive numbering after switch fusing)
    bus_type: np.ndarray          # int32 [nb]  PQ/PV/REF
    vn_kv: np.ndarray             # f64 [nb]
    vm_set: np.ndarray            # f64 [nb] set-point for PV/REF, 1.0 for PQ
    va_set: np.ndarray            # f64 [nb] radians; REF angle, else start angle
    gs: np.ndarray                # f64 [nb] shunt conductance p.u.
    bs: np.ndarray                # f64 [nb] shunt susceptance p.u.
    # branches: Ybus stamps  (makeYbus)
    f: np.ndarray                 # int32 [nbr]
    t: np.ndarray                 # int32 [nbr]
    yff: np.ndarray               # c128 [nbr]
    yft: np.ndarray
    ytf: np.ndarray
    ytt: np.ndarray
    kf: np.ndarray                # f64 [nbr] loading% = max(|If|*kf, |It|*kt)
    kt: np.ndarray
    br_kind: np.ndarray           # int32 [nbr]  0 line / 1 trafo / 2 trafo3w winding / 3 impedance / 4 bus-bus switch with z_ohm / 5 xward impedance
    br_elem: np.ndarray           # int32 [nbr]  positional row in net.line / net.trafo / net.trafo3w / net.impedance / net.switch / net.xward
    br_side: np.ndarray = None    # int32 [nbr]  trafo3w: 0 hv, 1 mv, 2 lv winding of its star equivalent (else 0)
    # DC model of the branches (pypower makeBdc; for opfx_solve_opts.init = OPFX_INIT_DC)
    bdc: np.ndarray = None        # f64 [nbr]  1 / (x * ratio), 0 for a branch that couples nothing (open-ended)
    pfinj: np.ndarray = None      # f64 [nbr]  bdc * (-shift in rad)
    # lookups back to the net
    bus_lookup: dict = field(default_factory=dict)   # net bus index -> case bus
    ref_elems: np.ndarray = None  # positional ext_grid rows per REF bus order
    meta: dict = field(default_factory=dict)

    @property
    def nb(self) -> int:
        return len(self.bus_type)

    @property
    def nbr(self) -> int:
        return len(self.f)

    def ybus_dense(self) -> np.ndarray:
        y = np.zeros((self.nb, self.nb), dtype=np.complex128)
        np.add.at(y, (self.f, self.f), self.yff)
        np.add.at(y, (self.f, self.t), self.yft)
        np.add.at(y, (self.t, self.f), self.ytf)
        np.add.at(y, (self.t, self.t), self.ytt)
        y[np.arange(self.nb), np.arange(self.nb)] += self.gs + 1j * self.bs
        return y


def branch_stamps(r, x, bc, ratio, shift_rad):
    """pypower makeYbus branch part: Ys = 1/(r+jx); Ytt = Ys + j*Bc/2;
    Yff = Ytt/(tap*conj(tap)); Yft = -Ys/conj(tap); Ytf = -Ys/tap.
    `bc` may be complex (pandapower stores transformer iron losses there)."""
    ys = 1.0 / (np.asarray(r, dtype=float) + 1j * np.asarray(x, dtype=float))
    tap = np.asarray(ratio, dtype=float) * np.exp(1j * np.asarray(shift_rad, dtype=float))
    ytt = ys + 1j * np.asarray(bc, dtype=complex) / 2.0
    yff = ytt / (tap * np.conj(tap))
    yft = -ys / np.conj(tap)
    ytf = -ys / tap
    return yff, yft, ytf, ytt


def open_ended_stamps(yff, yft, ytf, ytt, side):
    """Branch with an open switch at one end (side 1: from/hv end open, 2: to/lv end open): no current
    enters at the open end, so its voltage follows from I = 0 there and what remains is a shunt at
    the connected end — ytt - ytf yft / yff, resp. yff - yft ytf / ytt — exactly what pandapower's
    auxiliary bus at the open end gives."""
    yff, yft, ytf, ytt = (np.array(a, dtype=complex) for a in (yff, yft, ytf, ytt))
    side = np.asarray(side)
    o1, o2 = side == 1, side == 2
    with np.errstate(divide='ignore', invalid='ignore'):
        eq_t = ytt - ytf * yft / yff
        eq_f = yff - yft * ytf / ytt
    ytt = np.where(o1, eq_t, np.where(o2, 0.0, ytt))
    yff = np.where(o2, eq_f, np.where(o1, 0.0, yff))
    yft = np.where(o1 | o2, 0.0, yft)
    ytf = np.where(o1 | o2, 0.0, ytf)
    return yff, yft, ytf, ytt


class _UnionFind:
    def __init__(self, keys):
        self.p = {k: k for k in keys}

    def find(self, a):
        while self.p[a] != a:
            self.p[a] = self.p[self.p[a]]
            a = self.p[a]
        return a

    def union(self, a, b):
        ra, rb = self.find(a), self.find(b)
        if ra != rb:
            self.p[max(ra, rb)] = min(ra, rb)


def _col(df, name, default):
    if name in df.columns:
        v = df[name].to_numpy()
        try:
            v = v.astype(float)
            return np.where(np.isnan(v), default, v)
        except (ValueError, TypeError):
            return v
    return np.full(len(df), default)


def _flags(df, name, default=False):
    """Boolean column that may arrive as object dtype with None / NaN entries (a real pandapower table)."""
    if name not in df.columns:
        return np.full(len(df), default, dtype=bool)
    return np.array([default if (v is None or v != v) else bool(v) for v in df[name].to_numpy()], dtype=bool)


# pandapower element tables that take part in `runpp` and that this converter has no model for: a net that
# fills one of them would be solved WITHOUT those elements — a silently different grid — so it is refused
UNMODELLED_TABLES = ('asymmetric_load', 'asymmetric_sgen',
                     'svc', 'tcsc', 'ssc', 'vsc', 'b2b_vsc', 'bus_dc', 'line_dc')


def _table(net, name):
    """net[name] if the net has such a table with rows, else None (a pandapowerNet of another version may lack it)."""
    try:
        df = net[name]
    except (KeyError, AttributeError):
        return None
    return df if hasattr(df, 'columns') and len(df) else None


def _count_nonzero(df, col):
    if col not in df.columns:
        return 0
    v = pd_to_float(df[col].to_numpy())
    return int(np.count_nonzero(np.nan_to_num(v)))


def pd_to_float(values):
    try:
        return np.asarray(values, dtype=float)
    except (TypeError, ValueError):
        return np.array([np.nan if (v is None or isinstance(v, str)) else float(v) for v in values])


def expand_dclines(net):
    """pandapower runs a DC line as TWO GENERATORS (`_add_dcline_gens`, before `_pd2ppc`): appended to the generator table per
    dcline row, first one at the to bus feeding in p_mw (1 - loss_percent / 100) - loss_mw at vm_to_pu within
    [min_q_to_mvar, max_q_to_mvar], then one at the from bus taking p_mw at vm_from_pu within the from-side range.
    Returns `net` itself when it holds no dcline row (or is expanded already), else a shallow copy — same tables, a longer
    `gen` — marked with the number of rows added (`_dcline_gens`); the rows of the original generators keep their positions,
    so results by generator position are those of the net's own generators first, the lines' two per row after them."""
    dc = _table(net, 'dcline')
    if dc is None or '_dcline_gens' in net:
        return net
    import copy
    import pandas as pd
    gen = net['gen']
    on = _col(dc, 'in_service', True).astype(bool)
    loss = _col(dc, 'loss_percent', 0.0) / 100.0
    rows = []
    for pos in range(len(dc)):
        g = lambda c_, d=np.nan: float(_col(dc, c_, d)[pos])
        p_from = g('p_mw')
        p_to = p_from * (1.0 - loss[pos]) - g('loss_mw', 0.0)
        rows.append(dict(bus=int(dc['to_bus'].iloc[pos]), p_mw=p_to, vm_pu=g('vm_to_pu'), min_q_mvar=g('min_q_to_mvar'),
                         max_q_mvar=g('max_q_to_mvar'), scaling=1.0, in_service=bool(on[pos]), controllable=False, name=None))
        rows.append(dict(bus=int(dc['from_bus'].iloc[pos]), p_mw=-p_from, vm_pu=g('vm_from_pu'), min_q_mvar=g('min_q_from_mvar'),
                         max_q_mvar=g('max_q_from_mvar'), scaling=1.0, in_service=bool(on[pos]), controllable=False, name=None))
    first = (int(gen.index.max()) + 1) if len(gen) else 0
    add = pd.DataFrame(rows, index=range(first, first + len(rows)))
    for c_ in gen.columns:                  # (columns of the net's own generators that a line's generators do not have)
        if c_ not in add.columns:
            add[c_] = False if gen[c_].dtype == bool else np.nan
    out = copy.copy(net)
    out['gen'] = pd.concat([gen, add[list(gen.columns) + [c_ for c_ in add.columns if c_ not in gen.columns]]]) if len(gen) else add
    out['gen']['bus'] = out['gen']['bus'].astype(np.int64)
    out['gen']['in_service'] = out['gen']['in_service'].astype(bool)
    out['_dcline_gens'] = len(rows)
    return out


def check_supported(net) -> None:
    """Raise ValueError (naming the table / column) for net content that `pp.runpp` would model and this
    converter does not (opf_env.py:703 hands the WHOLE net to pandapower): rows in the
    asymmetric / FACTS tables; voltage-dependent (ZIP) loads; line / transformer switches with an
    impedance; switches at three-winding transformers; generators acting as slack;
    characteristic-dependent transformer impedances."""
    def table(name):
        try:
            df = net[name]
        except (KeyError, AttributeError):
            return None
        return df if hasattr(df, 'columns') and len(df) else None
    for name in UNMODELLED_TABLES:
        if table(name) is not None:
            raise ValueError(f'net.{name} has {len(net[name])} row(s): this element type is not modelled by the '
                             f'batched power flow (pandapower would include it)')
    load = table('load')
    if load is not None:
        for col in ('const_z_percent', 'const_i_percent'):
            if _count_nonzero(load, col):
                raise ValueError(f'net.load.{col} is non-zero: voltage-dependent (ZIP) loads are not modelled')
    sw = table('switch')
    if sw is not None:
        if 'z_ohm' in sw.columns and np.nan_to_num(pd_to_float(sw['z_ohm'].to_numpy()))[[str(v) != 'b' for v in sw['et']]].any():
            raise ValueError('net.switch.z_ohm is non-zero at a line / transformer switch: only bus-bus switches with an '
                             'impedance are modelled')
        if 'et' in sw.columns and any(str(v) == 't3' for v in sw['et']):
            raise ValueError("net.switch: et='t3' (switches at three-winding transformers) is not modelled")
    gen = table('gen')
    if gen is not None and _flags(gen, 'slack').any():
        raise ValueError('net.gen.slack is set: generators acting as slack are not modelled (use an ext_grid)')
    for name in ('trafo', 'trafo3w'):
        tr = table(name)
        if tr is not None and _flags(tr, 'tap_dependent_impedance').any():
            raise ValueError(f'net.{name}.tap_dependent_impedance is set: characteristic-dependent transformer '
                             f'impedances are not modelled')
    t3 = table('trafo3w')
    if t3 is not None and _flags(t3, 'tap_at_star_point').any():
        raise ValueError('net.trafo3w.tap_at_star_point is set: not modelled')
    if t3 is not None and _count_nonzero(t3, 'tap_step_degree'):
        raise ValueError('net.trafo3w.tap_step_degree is non-zero: not modelled for three-winding transformers')


def tap_changer(side, tap_pos, tap_neutral, step_percent, step_degree, phase_shifter):
    """pandapower `_calc_tap_from_dataframe` for one transformer: (factor on the rated voltage of the tap
    side, additional phase shift in degree).  A ratio/asymmetrical tap changer moves the voltage phasor by
    du = step_percent/100 * (pos - neutral) under the angle `step_degree`: |1 + du e^{j a}| and
    arctan(+-du sin a / (1 + du cos a)) (+ on the hv side, - on the lv side); an ideal phase shifter
    (`tap_phase_shifter`) only turns the angle, by `step_degree` per step or, given as `step_percent`,
    by 2 asin(du / 2)."""
    if side not in ('hv', 'lv') or not np.isfinite(tap_pos):
        return 1.0, 0.0
    direction = 1.0 if side == 'hv' else -1.0
    diff = tap_pos - tap_neutral
    deg = step_degree if np.isfinite(step_degree) else 0.0
    pct = step_percent if np.isfinite(step_percent) else 0.0
    if phase_shifter:
        if deg != 0.0 and pct != 0.0:
            raise ValueError('trafo: ideal phase shifter with both tap_step_degree and tap_step_percent')
        if deg != 0.0:
            return 1.0, direction * diff * deg
        return 1.0, direction * 2.0 * np.degrees(np.arcsin(diff * pct / 100.0 / 2.0))
    if not np.isfinite(step_percent):
        return 1.0, 0.0
    du = step_percent * diff / 100.0
    if not np.isfinite(du):                  # (no neutral position given: pandapower applies no tap)
        return 1.0, 0.0
    a = np.radians(deg)
    re, im = 1.0 + du * np.cos(a), du * np.sin(a)
    return float(np.hypot(re, im)), float(np.degrees(np.arctan(direction * im / re)))


def net_to_case(net, calculate_voltage_angles='auto') -> Case:
    """Convert the element tables of `net` into a per-unit :class:`Case`.

    Restates pandapower `_pd2ppc` (third party):
      * closed bus-bus switches fuse buses; a line / transformer with an open
        switch at BOTH ends is out of service; with an open switch at ONE end
        it stays connected at the other (pandapower re-routes the open end to
        an auxiliary bus, so the element keeps drawing its charging /
        magnetising current): here the auxiliary bus is eliminated exactly —
        the branch becomes a shunt y = Yff - Yft Ytf / Ytt at its connected
        end (`open_ended_stamps`), with no coupling to the other bus;
      * line:  r,x [Ω/km]·len/parallel ÷ Zbase,  b = 2πf·C·len·parallel·Zbase,
        Zbase = vn_kv(from bus)² / sn_mva;
      * trafo: short-circuit impedance from vk/vkr referred to the LV side,
        magnetising branch from pfe/i0, T-model converted to π by a wye-delta
        transform (`trafo_model='t'`), off-nominal ratio from the tap changer,
        phase shift when voltage angles are calculated;
      * every in-service ext_grid bus is REF, every in-service gen bus PV.
    """
    check_supported(net)
    net = expand_dclines(net)
    base = float(net['sn_mva']) if 'sn_mva' in net else 1.0
    f_hz = float(net['f_hz']) if 'f_hz' in net else 50.0
    bus_df = net['bus']
    in_service_bus = _col(bus_df, 'in_service', True).astype(bool)
    bus_ids = list(bus_df.index)

    # --- switches ---------------------------------------------------------
    uf = _UnionFind(bus_ids)
    line_open, trafo_open = {}, {}          # element -> buses at which one of its switches is open
    sw = net['switch'] if 'switch' in net else None
    z_switches = []                         # closed bus-bus switches with an impedance: (row, bus, other bus, z_ohm)
    if sw is not None and len(sw):
        z_sw = np.nan_to_num(pd_to_float(sw['z_ohm'].to_numpy())) if 'z_ohm' in sw.columns else np.zeros(len(sw))
        for pos, (b, e, et, closed) in enumerate(zip(sw['bus'], sw['element'], sw['et'], sw['closed'])):
            if et == 'b':
                if closed and z_sw[pos] > 0:
                    z_switches.append((pos, int(b), int(e), float(z_sw[pos])))
                elif closed:
                    uf.union(int(b), int(e))
            elif et == 'l':
                if not closed:
                    line_open.setdefault(int(e), set()).add(int(b))
            elif et == 't':
                if not closed:
                    trafo_open.setdefault(int(e), set()).add(int(b))

    def open_side(opened, elem, fb, tb):
        """0: connected at both ends, 1: open at the from/hv end, 2: at the to/lv end, 3: both"""
        at = opened.get(elem, ())
        return (1 if fb in at else 0) | (2 if tb in at else 0)

    # --- ext_grid / gen decide angle handling -------------------------------
    eg = net['ext_grid']
    eg_on = _col(eg, 'in_service', True).astype(bool) if len(eg) else np.zeros(0, bool)
    if calculate_voltage_angles == 'auto':
        # pandapower: angles are calculated iff an ext_grid sits above 70 kV
        calc_angles = bool(len(eg)) and bool(
            (bus_df.loc[eg['bus'].to_numpy()[eg_on], 'vn_kv'].to_numpy(float) > 70.0).any())
    else:
        calc_angles = bool(calculate_voltage_angles)

    # --- branches -----------------------------------------------------------
    ln = net['line']
    tr = net['trafo']
    rows = []  # (root_f, root_t, r, x, bc, ratio, shift, kind, elem_pos, kf_num, kt_num, open side)
    vn = bus_df['vn_kv'].astype(float)
    bus_pos = {int(b): i for i, b in enumerate(bus_ids)}
    bus_on = lambda b: in_service_bus[bus_pos[b]]
    if len(ln):
        on = _col(ln, 'in_service', True).astype(bool)
        par = _col(ln, 'parallel', 1.0)
        dfac = _col(ln, 'df', 1.0)
        g_us = _col(ln, 'g_us_per_km', 0.0)
        num = lambda c: ln[c].to_numpy(float)
        fbs, tbs = ln['from_bus'].to_numpy().astype(np.int64), ln['to_bus'].to_numpy().astype(np.int64)
        vn_f, vn_t = vn.reindex(fbs).to_numpy(), vn.reindex(tbs).to_numpy()
        length = num('length_km')
        zb = vn_f ** 2 / base
        r_pu = num('r_ohm_per_km') * length / par / zb
        x_pu = num('x_ohm_per_km') * length / par / zb
        b_pu = 2 * np.pi * f_hz * num('c_nf_per_km') * 1e-9 * length * par * zb
        g_pu = g_us * 1e-6 * length * par * zb
        imax = num('max_i_ka') * dfac * par
        with np.errstate(divide='ignore', invalid='ignore'):
            kf_l = base / (np.sqrt(3.0) * vn_f * imax) * 100.0
            kt_l = base / (np.sqrt(3.0) * vn_t * imax) * 100.0
        for pos, idx in enumerate(ln.index):
            fb, tb = int(fbs[pos]), int(tbs[pos])
            side = open_side(line_open, int(idx), fb, tb)
            if not on[pos] or side == 3:
                continue
            if not (bus_on(fb) and bus_on(tb)):
                continue
            bc = b_pu[pos] - 1j * g_pu[pos]      # j*bc/2 = (g + jb)/2 per side
            rows.append((fb, tb, r_pu[pos], x_pu[pos], bc, 1.0, 0.0, KIND_LINE, pos, kf_l[pos], kt_l[pos], side, 0))
    if len(tr):
        on = _col(tr, 'in_service', True).astype(bool)
        par = _col(tr, 'parallel', 1.0)
        dfac = _col(tr, 'df', 1.0)
        tap_pos = _col(tr, 'tap_pos', np.nan)
        tap_neutral = _col(tr, 'tap_neutral', np.nan)
        tap_step = _col(tr, 'tap_step_percent', np.nan)
        tap_deg = _col(tr, 'tap_step_degree', np.nan)
        phase_shifter = _flags(tr, 'tap_phase_shifter')
        shift = _col(tr, 'shift_degree', 0.0)
        for pos, idx in enumerate(tr.index):
            hb, lb = int(tr.at[idx, 'hv_bus']), int(tr.at[idx, 'lv_bus'])
            oside = open_side(trafo_open, int(idx), hb, lb)
            if not on[pos] or oside == 3:
                continue
            if not (bus_on(hb) and bus_on(lb)):
                continue
            sn = float(tr.at[idx, 'sn_mva'])
            vn_hv, vn_lv = float(tr.at[idx, 'vn_hv_kv']), float(tr.at[idx, 'vn_lv_kv'])
            vt_hv, vt_lv = vn_hv, vn_lv
            side = tr.at[idx, 'tap_side'] if 'tap_side' in tr.columns else None
            fac, tap_shift = tap_changer(side if isinstance(side, str) else '', tap_pos[pos], tap_neutral[pos],
                                         tap_step[pos], tap_deg[pos], phase_shifter[pos])
            if side == 'hv':
                vt_hv = vn_hv * fac
            elif side == 'lv':
                vt_lv = vn_lv * fac
            vb_hv, vb_lv = vn[hb], vn[lb]
            # short-circuit impedance referred to the LV side, system base
            tap_lv = (vt_lv / vb_lv) ** 2 * base
            z_sc = float(tr.at[idx, 'vk_percent']) / 100.0 / sn * tap_lv
            r_sc = float(tr.at[idx, 'vkr_percent']) / 100.0 / sn * tap_lv
            x_sc = np.sign(z_sc) * np.sqrt(max(z_sc ** 2 - r_sc ** 2, 0.0))
            r_sc /= par[pos]
            x_sc /= par[pos]
            # magnetising admittance
            base_r = vb_lv ** 2 / base
            pfe = float(tr.at[idx, 'pfe_kw']) * 1e-3
            vnl2 = vn_lv ** 2
            b_real = pfe / vnl2 * base_r
            i0 = float(tr.at[idx, 'i0_percent'])
            b_img2 = (i0 / 100.0 * sn) ** 2 - pfe ** 2
            b_img = np.sqrt(max(b_img2, 0.0)) * base_r / vnl2
            y = (-1j * b_real - b_img * np.sign(i0)) / (vt_lv / vn_lv) ** 2 * par[pos]
            # T -> pi (wye-delta), only when a magnetising branch exists
            if y != 0:
                za = zb_ = (r_sc + 1j * x_sc) / 2.0
                zc = -1j / y
                zsum = za * zb_ + za * zc + zb_ * zc
                zab = zsum / zc
                zac = zsum / zb_
                r_pi, x_pi = zab.real, zab.imag
                bc = -2j / zac
            else:
                r_pi, x_pi, bc = r_sc, x_sc, 0.0 + 0.0j
            ratio = (vt_hv / vb_hv) / (vt_lv / vb_lv)
            # (shift_degree counts only when angles are calculated; the tap changer's own shift always does)
            sh = np.deg2rad((shift[pos] if calc_angles else 0.0) + tap_shift)
            kf = base * (vn_hv / vb_hv) / sn * 100.0 / (par[pos] * dfac[pos])
            kt = base * (vn_lv / vb_lv) / sn * 100.0 / (par[pos] * dfac[pos])
            rows.append((hb, lb, r_pi, x_pi, bc, ratio, sh, KIND_TRAFO, pos, kf, kt, oside, 0))

    # --- series impedances (pandapower `_calc_impedance_parameter`): per unit of their own sn_mva, referred to the net's;
    # the to-side values may differ: Yff = -Yft = 1 / z_ft, Ytt = -Ytf = 1 / z_tf (pandapower's makeYbus, BR_R_ASYM / BR_X_ASYM)
    asym = {}                                          # (kind, row) -> (r, x) seen from the to side
    imp = _table(net, 'impedance')
    if imp is not None:
        on = _col(imp, 'in_service', True).astype(bool)
        k_sn = base / imp['sn_mva'].to_numpy(float)
        num = lambda c_: imp[c_].to_numpy(float)
        for pos, idx in enumerate(imp.index):
            fb, tb = int(imp.at[idx, 'from_bus']), int(imp.at[idx, 'to_bus'])
            if not on[pos] or not (bus_on(fb) and bus_on(tb)):
                continue
            rows.append((fb, tb, num('rft_pu')[pos] * k_sn[pos], num('xft_pu')[pos] * k_sn[pos], 0.0 + 0.0j, 1.0, 0.0,
                         KIND_IMPEDANCE, pos, 0.0, 0.0, 0, 0))
            asym[(KIND_IMPEDANCE, pos)] = (num('rtf_pu')[pos] * k_sn[pos], num('xtf_pu')[pos] * k_sn[pos])
    # --- closed bus-bus switches with an impedance (pandapower `_calc_switch_parameter`): |z| = z_ohm, r / x = switch_rx_ratio
    for pos, b, e, z in z_switches:
        if not (bus_on(b) and bus_on(e)):
            continue
        zb = float(vn[b]) ** 2 / base
        rows.append((b, e, z * SWITCH_RX_RATIO / np.hypot(1.0, SWITCH_RX_RATIO) / zb, z / np.hypot(1.0, SWITCH_RX_RATIO) / zb,
                     0.0 + 0.0j, 1.0, 0.0, KIND_SWITCH, pos, 0.0, 0.0, 0, 0))

    # --- three-winding transformers: star equivalent (pandapower `_trafo_df_from_trafo3w`) -------------
    # an auxiliary bus on the hv voltage level per transformer and three two-winding transformers
    # hv-bus -> star, star -> mv-bus, star -> lv-bus; the pairwise short-circuit voltages (hv-mv, mv-lv,
    # hv-lv, each on the smaller rating of its pair) are referred to sn_hv and turned from delta into wye
    # values; iron losses and magnetising current sit on the hv winding (`trafo3w_losses='hv'`)
    t3 = net['trafo3w'] if 'trafo3w' in net else None
    aux_vn = []                                        # one auxiliary bus per (in-service) trafo3w: ids -1, -2, ...
    if t3 is not None and len(t3):
        on3 = _col(t3, 'in_service', True).astype(bool)
        g = lambda c_, d=np.nan: _col(t3, c_, d)
        sn3 = np.stack([g('sn_hv_mva'), g('sn_mv_mva'), g('sn_lv_mva')])
        def on_hv(z):
            return sn3[0] * np.array([z[0] / np.minimum(sn3[0], sn3[1]), z[1] / np.minimum(sn3[1], sn3[2]),
                                      z[2] / np.minimum(sn3[0], sn3[2])])
        def delta_to_wye(z):
            return 0.5 * sn3 / sn3[0] * np.array([z[0] + z[2] - z[1], z[1] + z[0] - z[2], z[2] + z[1] - z[0]])
        vk_d = on_hv(np.stack([g('vk_hv_percent', 0.0), g('vk_mv_percent', 0.0), g('vk_lv_percent', 0.0)]))
        vkr_d = on_hv(np.stack([g('vkr_hv_percent', 0.0), g('vkr_mv_percent', 0.0), g('vkr_lv_percent', 0.0)]))
        vki_w = delta_to_wye(np.sqrt(vk_d ** 2 - vkr_d ** 2))
        vkr_w = delta_to_wye(vkr_d)
        vk_w = np.sign(vki_w) * np.sqrt(vki_w ** 2 + vkr_w ** 2)
        tap_pos3, tap_neu3, tap_stp3 = g('tap_pos'), g('tap_neutral', 0.0), g('tap_step_percent')
        for pos, idx in enumerate(t3.index):
            ends = [int(t3.at[idx, c_]) for c_ in ('hv_bus', 'mv_bus', 'lv_bus')]
            if not on3[pos] or not all(bus_on(b) for b in ends):
                continue
            if (vk_w[:, pos] == 0).any():
                raise ValueError('trafo3w: equivalent transformer with zero impedance')
            star = -1 - len(aux_vn)
            vn_h = float(t3.at[idx, 'vn_hv_kv'])
            aux_vn.append(vn_h)
            tside = t3.at[idx, 'tap_side'] if 'tap_side' in t3.columns else None
            fac = 1.0
            if isinstance(tside, str) and not np.isnan(tap_pos3[pos]) and not np.isnan(tap_stp3[pos]):
                fac = 1.0 + (tap_pos3[pos] - tap_neu3[pos]) * tap_stp3[pos] / 100.0
            for w, wname in enumerate(('hv', 'mv', 'lv')):
                sn_w = float(sn3[w, pos])
                vn_w = float(t3.at[idx, f'vn_{wname}_kv'])
                if w == 0:          # hv bus -> star: rated vn_hv / vn_hv, tap on the hv side
                    fb, tb, vb_f, vb_t = ends[0], star, float(vn[ends[0]]), vn_h
                    vt_f, vt_t, rated_t = vn_h * (fac if tside == 'hv' else 1.0), vn_h, vn_h
                else:               # star -> mv / lv bus: rated vn_hv / vn_side, tap on that side
                    fb, tb, vb_f, vb_t = star, ends[w], vn_h, float(vn[ends[w]])
                    vt_f, vt_t, rated_t = vn_h, vn_w * (fac if tside == wname else 1.0), vn_w
                tap_lv = (vt_t / vb_t) ** 2 * base
                z_sc = vk_w[w, pos] / 100.0 / sn_w * tap_lv
                r_sc = vkr_w[w, pos] / 100.0 / sn_w * tap_lv
                x_sc = np.sign(z_sc) * np.sqrt(max(z_sc ** 2 - r_sc ** 2, 0.0))
                r_pi, x_pi, bc = r_sc, x_sc, 0.0 + 0.0j
                pfe = float(g('pfe_kw', 0.0)[pos]) * 1e-3 if w == 0 else 0.0
                i0 = float(g('i0_percent', 0.0)[pos]) if w == 0 else 0.0
                if pfe or i0:
                    base_r = vb_t ** 2 / base
                    b_real = pfe / rated_t ** 2 * base_r
                    b_img = np.sqrt(max((i0 / 100.0 * sn_w) ** 2 - pfe ** 2, 0.0)) * base_r / rated_t ** 2
                    y = (-1j * b_real - b_img * np.sign(i0)) / (vt_t / rated_t) ** 2
                    if y != 0:
                        za = (r_sc + 1j * x_sc) / 2.0
                        zc = -1j / y
                        zsum = za * za + 2 * za * zc
                        r_pi, x_pi, bc = (zsum / zc).real, (zsum / zc).imag, -2j / (zsum / za)
                ratio = (vt_f / vt_t) / (vb_f / vb_t)
                sh = np.deg2rad(float(g(f'shift_{wname}_degree', 0.0)[pos])) if (calc_angles and w) else 0.0
                # loading: the current at the winding's own terminal (hv winding: from end, mv / lv: to end)
                k_term = base * (vn_w / (vb_f if w == 0 else vb_t)) / sn_w * 100.0
                rows.append((fb, tb, r_pi, x_pi, bc, ratio, sh, KIND_TRAFO3W, pos, k_term if w == 0 else 0.0,
                             0.0 if w == 0 else k_term, 0, w))

    # --- extended wards (pandapower `_calc_xward_parameter`, `_build_gen_ppc`): the ward's two parts at the bus, and behind
    # r_ohm + j x_ohm an internal bus that a source without active power holds at vm_pu — an auxiliary PV bus per xward
    xw = _table(net, 'xward')
    xward_aux = {}                                     # row -> (auxiliary bus id, vm_pu)
    if xw is not None:
        on = _col(xw, 'in_service', True).astype(bool)
        for pos, idx in enumerate(xw.index):
            b = int(xw.at[idx, 'bus'])
            if not on[pos] or not bus_on(b):
                continue
            aux = -1 - len(aux_vn)
            aux_vn.append(float(vn[b]))
            xward_aux[pos] = (aux, float(xw.at[idx, 'vm_pu']))
            zb = float(vn[b]) ** 2 / base
            rows.append((b, aux, float(xw.at[idx, 'r_ohm']) / zb, float(xw.at[idx, 'x_ohm']) / zb, 0.0 + 0.0j, 1.0, 0.0,
                         KIND_XWARD, pos, 0.0, 0.0, 0, 0))

    # --- which fused buses are alive: connected to a REF through branches ----
    roots = {b: uf.find(b) for b in bus_ids}
    for a_ in range(len(aux_vn)):
        roots[-1 - a_] = -1 - a_                   # auxiliary star buses of three-winding transformers
    eg_bus = [int(b) for b, on_ in zip(eg['bus'], eg_on) if on_] if len(eg) else []
    if not eg_bus:
        raise ValueError('net has no in-service ext_grid (no slack bus)')
    adj = {}
    for row in rows:
        if row[11]:                       # open-ended: hangs on one bus, connects nothing
            continue
        a, b = roots[row[0]], roots[row[1]]
        adj.setdefault(a, set()).add(b)
        adj.setdefault(b, set()).add(a)
    alive = set()
    stack = [roots[b] for b in eg_bus]
    while stack:
        a = stack.pop()
        if a in alive:
            continue
        alive.add(a)
        stack.extend(adj.get(a, ()))
    live_roots = {roots[b] for i, b in enumerate(bus_ids) if in_service_bus[i]} | {-1 - a_ for a_ in range(len(aux_vn))}
    alive &= live_roots
    order = sorted(r for r in alive if r >= 0) + sorted((r for r in alive if r < 0), reverse=True)   # auxiliary buses last
    root_to_case = {r: i for i, r in enumerate(order)}
    bus_lookup = {b: root_to_case[roots[b]] for i, b in enumerate(bus_ids)
                  if roots[b] in root_to_case and in_service_bus[i]}
    nb = len(order)

    def energised(r):
        near = [roots[r[0]] in root_to_case, roots[r[1]] in root_to_case]
        return all(near) if not r[11] else near[2 - r[11]]       # open-ended: its connected end must be alive
    rows = [r for r in rows if energised(r)]

    def end(r, which):
        """case bus of an end; the open end of an open-ended branch couples to nothing, so when its bus
        is not part of the case any other bus serves as the (unused) second index"""
        root = roots[r[which]]
        if root in root_to_case:
            return root_to_case[root]
        other = root_to_case[roots[r[1 - which]]]
        return (other + 1) % nb
    fcase = np.array([end(r, 0) for r in rows], dtype=np.int32)
    tcase = np.array([end(r, 1) for r in rows], dtype=np.int32)
    yff, yft, ytf, ytt = branch_stamps(
        [r[2] for r in rows], [r[3] for r in rows], [r[4] for r in rows],
        [r[5] for r in rows], [r[6] for r in rows])
    for k, r in enumerate(rows):
        if (r[7], r[8]) in asym:
            y_t = 1.0 / complex(*asym[(r[7], r[8])])
            ytt[k], ytf[k] = y_t, -y_t
    yff, yft, ytf, ytt = open_ended_stamps(yff, yft, ytf, ytt, np.array([r[11] for r in rows], dtype=np.int32))

    bus_type = np.full(nb, PQ, dtype=np.int32)
    vm_set = np.ones(nb)
    va_set = np.zeros(nb)
    vn_case = np.array([float(vn[r]) if r >= 0 else aux_vn[-1 - r] for r in order])
    gen = net['gen']
    if len(gen):
        g_on = _col(gen, 'in_service', True).astype(bool)
        for b, vm, on_ in zip(gen['bus'], gen['vm_pu'], g_on):
            if on_ and int(b) in bus_lookup:
                bus_type[bus_lookup[int(b)]] = PV
                vm_set[bus_lookup[int(b)]] = float(vm)
    xward_bus = {}
    for pos, (aux, vm) in xward_aux.items():
        if aux in root_to_case:
            i = root_to_case[aux]
            bus_type[i], vm_set[i] = PV, vm
            xward_bus[pos] = i
    ref_elems = []
    for pos, (b, on_) in enumerate(zip(eg['bus'], eg_on)):
        if on_ and int(b) in bus_lookup:
            i = bus_lookup[int(b)]
            bus_type[i] = REF
            vm_set[i] = float(eg['vm_pu'].iloc[pos])
            va_deg = float(eg['va_degree'].iloc[pos]) if 'va_degree' in eg.columns else 0.0
            va_set[i] = np.deg2rad(va_deg) if calc_angles else 0.0
            ref_elems.append(pos)

    gs = np.zeros(nb)
    bs = np.zeros(nb)
    sh_df = net['shunt'] if 'shunt' in net else None
    if sh_df is not None and len(sh_df):
        s_on = _col(sh_df, 'in_service', True).astype(bool)
        step = _col(sh_df, 'step', 1.0)
        for pos, b in enumerate(sh_df['bus']):
            if s_on[pos] and int(b) in bus_lookup:
                i = bus_lookup[int(b)]
                v_ratio = (vn[int(b)] / float(sh_df['vn_kv'].iloc[pos])) ** 2
                gs[i] += float(sh_df['p_mw'].iloc[pos]) * step[pos] * v_ratio / base
                bs[i] -= float(sh_df['q_mvar'].iloc[pos]) * step[pos] * v_ratio / base

    for name in ('ward', 'xward'):         # constant-impedance part of a ward: MW / Mvar at 1 p.u. (no voltage-level ratio)
        ward = _table(net, name)
        if ward is None:
            continue
        w_on = _col(ward, 'in_service', True).astype(bool)
        for pos, b in enumerate(ward['bus']):
            if w_on[pos] and int(b) in bus_lookup:
                gs[bus_lookup[int(b)]] += float(ward['pz_mw'].iloc[pos]) / base
                bs[bus_lookup[int(b)]] -= float(ward['qz_mvar'].iloc[pos]) / base

    # start angles: propagate the REF angle through transformer phase shifts
    # (equivalent in effect to pandapower's init='dc' for shifted MV grids:
    # the converged solution does not depend on the start)
    conn = np.array([r[11] == 0 for r in rows], dtype=bool)
    va0 = _propagate_angles(nb, fcase[conn], tcase[conn], [r[6] for r in rows if r[11] == 0], bus_type, va_set)
    va_set = np.where(bus_type == REF, va_set, va0)

    return Case(
        base_mva=base, bus_type=bus_type, vn_kv=vn_case, vm_set=vm_set,
        va_set=va_set, gs=gs, bs=bs, f=fcase, t=tcase, yff=yff, yft=yft,
        ytf=ytf, ytt=ytt,
        kf=np.array([r[9] for r in rows], dtype=float),
        kt=np.array([r[10] for r in rows], dtype=float),
        br_kind=np.array([r[7] for r in rows], dtype=np.int32),
        br_elem=np.array([r[8] for r in rows], dtype=np.int32),
        br_side=np.array([r[12] for r in rows], dtype=np.int32),
        bdc=_bdc(rows)[0], pfinj=_bdc(rows)[1],
        bus_lookup=bus_lookup, ref_elems=np.array(ref_elems, dtype=np.int32),
        meta={'calc_angles': calc_angles, 'xward_bus': xward_bus})


def _bdc(rows):
    """pypower `makeBdc` per branch: b = 1 / x, divided by the off-nominal ratio; phase-shift injection b * (-shift)."""
    x = np.array([r[3] for r in rows], dtype=float)
    ratio = np.array([r[5] for r in rows], dtype=float)
    shift = np.array([r[6] for r in rows], dtype=float)
    coupled = np.array([r[11] == 0 for r in rows], dtype=bool)
    with np.errstate(divide='ignore', invalid='ignore'):
        b = np.where(coupled, 1.0 / x / ratio, 0.0)
    return b, b * (-shift)


def _propagate_angles(nb, f, t, shift, bus_type, va_set):
    va0 = np.zeros(nb)
    seen = np.zeros(nb, bool)
    nbrs = [[] for _ in range(nb)]
    for k in range(len(f)):
        nbrs[f[k]].append((t[k], -shift[k]))
        nbrs[t[k]].append((f[k], +shift[k]))
    stack = []
    for i in np.flatnonzero(bus_type == REF):
        va0[i] = va_set[i]
        seen[i] = True
        stack.append(i)
    while stack:
        a = stack.pop()
        for b, d in nbrs[a]:
            if not seen[b]:
                seen[b] = True
                va0[b] = va0[a] + d
                stack.append(b)
    return va0


def bus_injections(net, case: Case):
    """makeSbus on the element tables: net injection per case
    bus in MW/Mvar (generation − demand) from load/sgen/storage/gen rows
    (`p_mw·scaling`, in service only), plus the summed reactive capability of
    the generators per bus.  Returns (p, q, qg_min, qg_max); q excludes the
    voltage-controlling generators (their Q is a result)."""
    net = expand_dclines(net)
    nb = case.nb
    p = np.zeros(nb)
    q = np.zeros(nb)
    for tbl, sign in (('load', -1.0), ('sgen', 1.0), ('storage', -1.0)):
        df = net[tbl]
        if not len(df):
            continue
        on = _col(df, 'in_service', True).astype(bool)
        sc = _col(df, 'scaling', 1.0)
        pv, qv = df['p_mw'].to_numpy(float), df['q_mvar'].to_numpy(float)
        for pos, b in enumerate(df['bus'].to_numpy()):
            if on[pos] and int(b) in case.bus_lookup:
                i = case.bus_lookup[int(b)]
                p[i] += sign * pv[pos] * sc[pos]
                q[i] += sign * qv[pos] * sc[pos]
    for tbl, (pv, qv) in static_consumption(net).items():
        for pos, b in enumerate(net[tbl]['bus'].to_numpy()):
            if int(b) in case.bus_lookup:
                p[case.bus_lookup[int(b)]] -= pv[pos]
                q[case.bus_lookup[int(b)]] -= qv[pos]
    qmin = np.full(nb, -np.inf)
    qmax = np.full(nb, np.inf)
    gen = net['gen']
    if len(gen):
        on = _col(gen, 'in_service', True).astype(bool)
        sc = _col(gen, 'scaling', 1.0)
        lo_c = _col(gen, 'min_q_mvar', -np.inf)
        hi_c = _col(gen, 'max_q_mvar', np.inf)
        acc_lo, acc_hi, has = np.zeros(nb), np.zeros(nb), np.zeros(nb, bool)
        for pos, b in enumerate(gen['bus'].to_numpy()):
            if on[pos] and int(b) in case.bus_lookup:
                i = case.bus_lookup[int(b)]
                p[i] += float(gen['p_mw'].iloc[pos]) * sc[pos]
                acc_lo[i] += lo_c[pos]
                acc_hi[i] += hi_c[pos]
                has[i] = True
        qmin[has], qmax[has] = acc_lo[has], acc_hi[has]
    return p, q, qmin, qmax


def static_consumption(net) -> dict:
    """Constant-power consumption of the element types that neither the reference's sampling nor its actuators touch, per
    row of their table in MW / Mvar (zero for a row out of service): {'ward': (p, q), 'xward': (p, q), 'motor': (p, q)}.

      * ward, xward (pandapower `_calc_pq_elements_and_add_on_ppc`): ps_mw, qs_mvar (the constant-impedance part is a shunt: net_to_case);
      * motor (pandapower `_get_motor_pq`): P = pn_mech_mw / (efficiency_percent / 100) * loading_percent / 100 * scaling,
        S = P / cos_phi, Q = sqrt(S^2 - P^2)."""
    out = {}
    for name in ('ward', 'xward'):
        ward = _table(net, name)
        if ward is not None:
            on = _col(ward, 'in_service', True).astype(float)
            out[name] = (ward['ps_mw'].to_numpy(float) * on, ward['qs_mvar'].to_numpy(float) * on)
    motor = _table(net, 'motor')
    if motor is not None:
        on = _col(motor, 'in_service', True).astype(float)
        p_m = (motor['pn_mech_mw'].to_numpy(float) / _col(motor, 'efficiency_percent', 100.0) * 100.0
               * _col(motor, 'loading_percent', 100.0) / 100.0 * _col(motor, 'scaling', 1.0) * on)
        s_m = p_m / motor['cos_phi'].to_numpy(float)
        out['motor'] = (p_m, np.sqrt(np.maximum(s_m ** 2 - p_m ** 2, 0.0)))
    return out


# pandapower's reactive limit of a generator row that names none (`_init_ppc_gen`: +-1e9 Mvar; ext_grids always) and the
# guard pypower's pfsoln adds to the summed range [3P]
Q_LIMIT_DEFAULT = 1e9
_PFSOLN_EPS = float(np.finfo(float).eps)


def generator_dispatch(net, case: Case) -> dict:
    """How the power generated at a bus is reported per GENERATOR (pypower `pfsoln` [3P]).

    The power flow knows generation per bus.  pypower's `pfsoln` — and with it `net.res_gen` / `net.res_ext_grid`, which
    the reference reads at objective.py:48-54 (cost rows with `et='gen'`) and opf_env.py:566-588 — writes it per
    generator row of the ppc (in-service ext_grids first, then in-service gens, pandapower `_build_gen_ppc`):

      * reactive power: the bus total Q_bus in equal shares where the summed range of the bus's generators is zero,
        otherwise in proportion to the ranges, Q_g = Qmin_g + (Q_bus - sum Qmin) / (sum Qmax - sum Qmin + eps) * (Qmax_g - Qmin_g);
        both are AFFINE in Q_bus: Q_g = q_a + q_b * Q_bus;
      * active power at a REF bus: the FIRST generator row of the bus balances it (what the solver reports as p_ext,
        the generators' own set-points already taken off), every other ext_grid there reports its ppc set-point, zero.

    Returns {'gen': {...}, 'ext_grid': {...}} with arrays over the rows of the two tables (the generator rows of DC lines, `expand_dclines`,
    after the net's own): `bus` (case bus, -1 for a row
    that takes no part in the power flow), `q_a` [Mvar], `q_b`, and for ext_grids `p_b` (1.0 / 0.0).  Q_bus is the
    solver's `q_gen[bus]` at PV buses (the limit total where `enforce_q_lims` has pinned the bus) and `q_ext` at REF buses.
    """
    net = expand_dclines(net)
    eg, gen = net['ext_grid'], net['gen']
    members = {}                                   # case bus -> [(table, pos, qmin, qmax)] in ppc row order
    eg_on = _col(eg, 'in_service', True).astype(bool) if len(eg) else np.zeros(0, bool)
    for pos, b in enumerate(eg['bus'].to_numpy() if len(eg) else ()):
        if eg_on[pos] and int(b) in case.bus_lookup:
            members.setdefault(case.bus_lookup[int(b)], []).append(('ext_grid', pos, -Q_LIMIT_DEFAULT, Q_LIMIT_DEFAULT))
    if len(gen):
        g_on = _col(gen, 'in_service', True).astype(bool)
        lo_c = _col(gen, 'min_q_mvar', -Q_LIMIT_DEFAULT)
        hi_c = _col(gen, 'max_q_mvar', Q_LIMIT_DEFAULT)
        for pos, b in enumerate(gen['bus'].to_numpy()):
            if g_on[pos] and int(b) in case.bus_lookup:
                members.setdefault(case.bus_lookup[int(b)], []).append(('gen', pos, float(lo_c[pos]), float(hi_c[pos])))
    out = {tbl: dict(bus=np.full(len(net[tbl]), -1, dtype=np.int64), q_a=np.zeros(len(net[tbl])), q_b=np.zeros(len(net[tbl])))
           for tbl in ('gen', 'ext_grid')}
    out['ext_grid']['p_b'] = np.zeros(len(eg))
    for bus, rows in members.items():
        lo_sum, hi_sum = sum(r[2] for r in rows), sum(r[3] for r in rows)
        for k, (tbl, pos, lo, hi) in enumerate(rows):
            o = out[tbl]
            o['bus'][pos] = bus
            if len(rows) == 1:
                o['q_a'][pos], o['q_b'][pos] = 0.0, 1.0
            elif lo_sum == hi_sum:
                o['q_a'][pos], o['q_b'][pos] = 0.0, 1.0 / len(rows)
            else:
                share = (hi - lo) / (hi_sum - lo_sum + _PFSOLN_EPS)
                o['q_a'][pos], o['q_b'][pos] = lo - lo_sum * share, share
            if tbl == 'ext_grid':
                o['p_b'][pos] = 1.0 if k == 0 else 0.0
    return out
