"""net -> per-unit power-flow case of the CPU oracle.

TEST INFRASTRUCTURE ONLY, and deliberately INDEPENDENT of the product: nothing
here imports the product package, and the product's own converter
(`case.py:net_to_case` there) shares no code with it.  The two are compared
against each other by `tests/test_pd2ppc_differential.py`.

What it restates: the third-party conversion that `pandapower.runpp` performs
before it solves (`_pd2ppc`; call site /root/reference/opfgym/opf_env.py:703;
pandapower `>=2.13.1,<3.0`, pyproject.toml:32, not vendored, not installed
here).  The structure follows pandapower's: a MATPOWER-shaped case with a bus
table, a branch table holding (r, x, b, tap, shift, status) — NOT admittance
stamps — and a generator table; closed bus-bus switches fuse buses; a line or
transformer behind ONE open switch stays connected at its other end and ends
at an auxiliary bus (it keeps drawing its charging / magnetising current);
open switches at both ends take it out of service; buses without a path to a
slack are isolated (`check_connectivity=True`).  Formulas restated from
pandapower's documentation of the element models ("Electric Model" of line,
trafo, trafo3w, shunt) and its published source, from memory: unverifiable
here — PARITY UNPINNED against pandapower itself (see pf_oracle.py).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

PQ, PV, REF, NONE = 1, 2, 3, 4


@dataclass
class PPC:
    """MATPOWER/pypower-shaped case (`ppc` in pandapower's terms)."""
    base_mva: float
    # ---- bus table [nb] --------------------------------------------------
    bus_type: np.ndarray      # PQ / PV / REF / NONE (isolated)
    pd: np.ndarray            # MW   demand (load + storage - sgen, times scaling)
    qd: np.ndarray            # MVAr
    gs: np.ndarray            # MW   shunt conductance at 1 p.u.
    bs: np.ndarray            # MVAr shunt susceptance at 1 p.u. (injected)
    vm: np.ndarray            # p.u. start / set-point magnitude
    va: np.ndarray            # degree
    base_kv: np.ndarray
    # ---- branch table [nbr] ------------------------------------------------
    f: np.ndarray
    t: np.ndarray
    r: np.ndarray             # p.u.
    x: np.ndarray
    b: np.ndarray             # complex: total charging susceptance b - j g  (as pandapower stores it)
    tap: np.ndarray           # off-nominal ratio (1 for lines)
    shift: np.ndarray         # degree
    status: np.ndarray        # 0 / 1
    br_table: list            # 'line' | 'trafo' | 'trafo3w'
    br_pos: np.ndarray        # positional row in that table
    br_side: list             # trafo3w: 'hv' | 'mv' | 'lv'; '' otherwise
    # ---- generator table [ng] ------------------------------------------------
    g_bus: np.ndarray
    g_p: np.ndarray           # MW (times scaling)
    g_qmin: np.ndarray        # MVAr
    g_qmax: np.ndarray
    g_vg: np.ndarray
    g_status: np.ndarray
    g_table: list             # 'ext_grid' | 'gen'
    g_pos: np.ndarray
    # ---- lookups ---------------------------------------------------------------
    bus_lookup: dict = field(default_factory=dict)     # net bus index -> ppc bus
    n_net_buses: int = 0                               # ppc buses >= this are auxiliary
    calc_angles: bool = False
    # ---- pandapower's extra branch columns BR_R_ASYM / BR_X_ASYM: what the to side sees more than the from side ------
    r_asym: np.ndarray = None
    x_asym: np.ndarray = None

    @property
    def nb(self):
        return len(self.bus_type)

    @property
    def nbr(self):
        return len(self.f)

    def branch_of(self, table, pos, side=''):
        for k in range(self.nbr):
            if self.br_table[k] == table and self.br_pos[k] == pos and self.br_side[k] == side:
                return k
        return -1


def _get(df, col, default):
    """Column as float array with NaN / missing replaced by `default`."""
    n = len(df)
    if col not in df.columns:
        return np.full(n, default, dtype=float)
    v = df[col].to_numpy()
    out = np.empty(n, dtype=float)
    for i, e in enumerate(v):
        try:
            fe = float(e)
        except (TypeError, ValueError):
            fe = np.nan
        out[i] = default if np.isnan(fe) else fe
    return out


def _flag(df, col, default=True):
    if col not in df.columns:
        return np.full(len(df), default, dtype=bool)
    return np.array([default if (v is None or v != v) else bool(v) for v in df[col].to_numpy()], dtype=bool)


class _Sets:
    def __init__(self, n):
        self.parent = list(range(n))

    def find(self, a):
        root = a
        while self.parent[root] != root:
            root = self.parent[root]
        while self.parent[a] != root:
            self.parent[a], a = root, self.parent[a]
        return root

    def join(self, a, b):
        ra, rb = self.find(a), self.find(b)
        if ra != rb:
            lo, hi = (ra, rb) if ra < rb else (rb, ra)
            self.parent[hi] = lo


def _line_parameters(net, base_mva, f_hz, vn_from):
    """pandapower `_calc_line_parameter`: r, x = ohm/km * km / parallel / Zbase with
    Zbase = vn(from bus)^2 / sn_mva; b = 2 pi f C(nF/km) 1e-9 km parallel Zbase; g likewise from
    g_us_per_km; the branch stores b - j g."""
    ln = net['line']
    length = _get(ln, 'length_km', 0.0)
    par = _get(ln, 'parallel', 1.0)
    zbase = vn_from ** 2 / base_mva
    r = _get(ln, 'r_ohm_per_km', 0.0) * length / zbase / par
    x = _get(ln, 'x_ohm_per_km', 0.0) * length / zbase / par
    b = 2.0 * np.pi * f_hz * _get(ln, 'c_nf_per_km', 0.0) * 1e-9 * zbase * length * par
    g = _get(ln, 'g_us_per_km', 0.0) * 1e-6 * zbase * length * par
    return r, x, b - 1j * g


def _tap_voltages(vn_hv, vn_lv, shift_deg, tap_side, tap_pos, tap_neutral, tap_step_percent, tap_step_degree,
                  phase_shifter, calc_angles):
    """pandapower `_calc_tap_from_dataframe`: rated voltages adjusted by the tap changer and the
    transformer's total phase shift (degree)."""
    vnh, vnl = vn_hv.copy(), vn_lv.copy()
    shift = shift_deg.copy() if calc_angles else np.zeros(len(vn_hv))
    diff = tap_pos - tap_neutral
    for side, vn, direction in (('hv', vnh, 1.0), ('lv', vnl, -1.0)):
        for k in range(len(vn)):
            if tap_side[k] != side or not np.isfinite(tap_pos[k]):
                continue
            if phase_shifter[k]:
                deg_set = np.isfinite(tap_step_degree[k]) and tap_step_degree[k] != 0
                pct_set = np.isfinite(tap_step_percent[k]) and tap_step_percent[k] != 0
                if deg_set and pct_set:
                    raise ValueError('ideal phase shifter with both tap_step_degree and tap_step_percent')
                if deg_set:
                    shift[k] += direction * diff[k] * tap_step_degree[k]
                elif pct_set:
                    shift[k] += direction * 2.0 * np.degrees(np.arcsin(diff[k] * tap_step_percent[k] / 100.0 / 2.0))
                continue
            if not np.isfinite(tap_step_percent[k]):
                continue
            step = tap_step_percent[k] * diff[k] / 100.0
            if not np.isfinite(step):            # `_replace_nan(tap_steps)`: no neutral position -> no tap change
                step = 0.0
            ang = np.radians(tap_step_degree[k]) if np.isfinite(tap_step_degree[k]) else 0.0
            u1 = vn[k]
            du = u1 * step
            vn[k] = np.sqrt((u1 + du * np.cos(ang)) ** 2 + (du * np.sin(ang)) ** 2)
            shift[k] += np.degrees(np.arctan(direction * du * np.sin(ang) / (u1 + du * np.cos(ang))))
    return vnh, vnl, shift


def _trafo_branch(base_mva, sn, vn_lv_rated, vn_lv_tap, vn_lv_bus, vk, vkr, pfe_kw, i0, parallel):
    """pandapower `_calc_r_x_y_from_dataframe` with `trafo_model='t'`: series impedance from the
    short-circuit voltages referred to the LV side, magnetising admittance from the open-circuit
    data, T equivalent converted to pi by a wye-delta transformation.  Returns r, x, b (complex)."""
    tap_lv = (vn_lv_tap / vn_lv_bus) ** 2 * base_mva
    z_sc = vk / 100.0 / sn * tap_lv
    r_sc = vkr / 100.0 / sn * tap_lv
    x_sc = np.sign(z_sc) * np.sqrt(np.maximum(z_sc ** 2 - r_sc ** 2, 0.0))
    r, x = r_sc / parallel, x_sc / parallel
    base_r = vn_lv_bus ** 2 / base_mva
    pfe = pfe_kw * 1e-3
    b_real = pfe / vn_lv_rated ** 2 * base_r
    b_img2 = (i0 / 100.0 * sn) ** 2 - pfe ** 2
    b_img = np.sqrt(np.maximum(b_img2, 0.0)) * base_r / vn_lv_rated ** 2
    y = (-1j * b_real - b_img * np.sign(i0)) / (vn_lv_tap / vn_lv_rated) ** 2 * parallel
    b = np.zeros(len(r), dtype=complex)
    for k in range(len(r)):
        if y[k] == 0:
            continue
        za = zb = (r[k] + 1j * x[k]) / 2.0
        zc = -1j / y[k]
        tri = za * zb + za * zc + zb * zc
        zab, zac = tri / zc, tri / zb
        r[k], x[k] = zab.real, zab.imag
        b[k] = -2j / zac
    return r, x, b


def _trafo3w_as_two_winding(t3):
    """pandapower `_trafo_df_from_trafo3w`: three two-winding transformers hv-star, star-mv, star-lv;
    short-circuit voltages: pairwise values referred to sn_hv, delta -> wye; open-circuit losses on the
    hv side (`trafo3w_losses='hv'`)."""
    n = len(t3)
    sn = np.stack([_get(t3, 'sn_hv_mva', np.nan), _get(t3, 'sn_mv_mva', np.nan), _get(t3, 'sn_lv_mva', np.nan)])
    vk3 = np.stack([_get(t3, 'vk_hv_percent', 0.0), _get(t3, 'vk_mv_percent', 0.0), _get(t3, 'vk_lv_percent', 0.0)])
    vkr3 = np.stack([_get(t3, 'vkr_hv_percent', 0.0), _get(t3, 'vkr_mv_percent', 0.0), _get(t3, 'vkr_lv_percent', 0.0)])

    def to_hv_rating(z):         # hv-mv, mv-lv, hv-lv values are given on the smaller rating of each pair
        return sn[0] * np.array([z[0] / np.minimum(sn[0], sn[1]), z[1] / np.minimum(sn[1], sn[2]),
                                 z[2] / np.minimum(sn[0], sn[2])])

    def wye(z):
        return 0.5 * sn / sn[0] * np.array([z[0] + z[2] - z[1], z[1] + z[0] - z[2], z[2] + z[1] - z[0]])
    vk_d, vkr_d = to_hv_rating(vk3), to_hv_rating(vkr3)
    vki_d = np.sqrt(vk_d ** 2 - vkr_d ** 2)
    vkr_w, vki_w = wye(vkr_d), wye(vki_d)
    vk_w = np.sign(vki_w) * np.sqrt(vki_w ** 2 + vkr_w ** 2)
    if (vk_w == 0).any():
        raise ValueError('equivalent transformer with zero impedance')
    zeros = np.zeros(n)
    out = {}
    for s, side in enumerate(('hv', 'mv', 'lv')):
        out[side] = dict(sn=sn[s], vk=vk_w[s], vkr=vkr_w[s],
                         pfe=_get(t3, 'pfe_kw', 0.0) if side == 'hv' else zeros,
                         i0=_get(t3, 'i0_percent', 0.0) if side == 'hv' else zeros,
                         vn_hv=_get(t3, 'vn_hv_kv', np.nan), vn_lv=_get(t3, f'vn_{side}_kv', np.nan),
                         shift=zeros if side == 'hv' else _get(t3, f'shift_{side}_degree', 0.0))
    return out


def motor_pq(mt):
    """pandapower `_get_motor_pq`: electrical power of the motors from their mechanical rating, MW / Mvar per row
    (in_service not applied)."""
    scale = _get(mt, 'loading_percent', 100.0) / 100.0 * _get(mt, 'scaling', 1.0)
    p_mw = _get(mt, 'pn_mech_mw', np.nan) / _get(mt, 'efficiency_percent', 100.0) * 100.0 * scale
    s_mva = p_mw / _get(mt, 'cos_phi', np.nan)
    return p_mw, np.sqrt(s_mva ** 2 - p_mw ** 2)


def refuse_unmodelled(net):
    """`pp.runpp` models every element table of the net; this restatement covers bus, line, trafo, trafo3w,
    load, sgen, storage, gen, ext_grid, shunt and switch.  Anything else that is filled in would make the
    oracle solve another grid than pandapower does, so it is an error, not a silent omission.  (Round 6: ward, xward, impedance,
    motor, dcline and closed bus-bus switches with z_ohm are covered too.)"""
    def rows(name):
        df = net[name] if name in net else None
        return df if df is not None and hasattr(df, 'columns') and len(df) else None

    def nonzero(df, col):
        return col in df.columns and bool(np.any(_get(df, col, 0.0) != 0.0))
    for name in ('asymmetric_load', 'asymmetric_sgen', 'svc',
                 'tcsc', 'ssc', 'vsc', 'b2b_vsc', 'bus_dc', 'line_dc'):
        if rows(name) is not None:
            raise ValueError(f'oracle: element table {name!r} is not modelled')
    ld = rows('load')
    if ld is not None and (nonzero(ld, 'const_z_percent') or nonzero(ld, 'const_i_percent')):
        raise ValueError('oracle: ZIP loads (load.const_z_percent / const_i_percent) are not modelled')
    sw = rows('switch')
    if sw is not None:
        if 'z_ohm' in sw.columns and any(z != 0.0 and str(e) != 'b' for z, e in zip(np.nan_to_num(_get(sw, 'z_ohm', 0.0)), sw['et'])):
            raise ValueError('oracle: switch.z_ohm at a line / transformer switch is not modelled')
        if any(str(e) == 't3' for e in sw['et']):
            raise ValueError("oracle: switch.et == 't3' is not modelled")
    gn = rows('gen')
    if gn is not None and _flag(gn, 'slack', False).any():
        raise ValueError('oracle: gen.slack is not modelled')
    for name in ('trafo', 'trafo3w'):
        tr = rows(name)
        if tr is not None and 'tap_dependent_impedance' in tr.columns and _flag(tr, 'tap_dependent_impedance', False).any():
            raise ValueError(f'oracle: {name}.tap_dependent_impedance is not modelled')
    t3 = rows('trafo3w')
    if t3 is not None and ((('tap_at_star_point' in t3.columns) and _flag(t3, 'tap_at_star_point', False).any())
                           or nonzero(t3, 'tap_step_degree')):
        raise ValueError('oracle: trafo3w.tap_at_star_point / tap_step_degree are not modelled')


def build_ppc(net, calculate_voltage_angles='auto', check_connectivity=True) -> PPC:
    refuse_unmodelled(net)
    base_mva = float(net['sn_mva']) if 'sn_mva' in net else 1.0
    f_hz = float(net['f_hz']) if 'f_hz' in net else 50.0
    bus_df = net['bus']
    n_bus = len(bus_df)
    pos_of = {int(b): i for i, b in enumerate(bus_df.index)}
    bus_on = _flag(bus_df, 'in_service')
    vn = bus_df['vn_kv'].to_numpy(float)

    # ---- bus fusing by closed bus-bus switches -------------------------------------
    sets = _Sets(n_bus)
    sw = net['switch'] if 'switch' in net and len(net['switch']) else None
    if sw is not None:
        sw_z = np.nan_to_num(_get(sw, 'z_ohm', 0.0))
        for b, e, et, closed, z in zip(sw['bus'], sw['element'], sw['et'], sw['closed'], sw_z):
            if et == 'b' and bool(closed) and int(b) in pos_of and int(e) in pos_of and not z > 0.0:     # (z_ohm > 0: a branch, below)
                sets.join(pos_of[int(b)], pos_of[int(e)])
    roots = sorted({sets.find(i) for i in range(n_bus)})
    root_id = {r: k for k, r in enumerate(roots)}
    lookup_pos = np.array([root_id[sets.find(i)] for i in range(n_bus)])
    bus_lookup = {int(b): int(lookup_pos[i]) for i, b in enumerate(bus_df.index)}
    nb = len(roots)
    base_kv = [vn[r] for r in roots]
    alive_bus = [any(bus_on[i] for i in range(n_bus) if lookup_pos[i] == k) for k in range(nb)]

    # ---- ext_grid / angle handling ----------------------------------------------------
    eg = net['ext_grid']
    eg_on = _flag(eg, 'in_service') if len(eg) else np.zeros(0, bool)
    if calculate_voltage_angles == 'auto':
        calc_angles = any(vn[pos_of[int(b)]] > 70.0 for b, on in zip(eg['bus'], eg_on) if on) if len(eg) else False
    else:
        calc_angles = bool(calculate_voltage_angles)

    f, t, r, x, b, tap, shift, status, tbl, bpos, bside, r_as, x_as = [], [], [], [], [], [], [], [], [], [], [], [], []

    def add_branch(fb, tb, r_, x_, b_, tap_, shift_, on, table, pos, side='', r_asym=0.0, x_asym=0.0):
        f.append(fb); t.append(tb); r.append(r_); x.append(x_); b.append(b_); tap.append(tap_)
        shift.append(shift_); status.append(1 if on else 0); tbl.append(table); bpos.append(pos); bside.append(side)
        r_as.append(r_asym); x_as.append(x_asym)

    # ---- lines ---------------------------------------------------------------------------
    ln = net['line']
    if len(ln):
        fb = np.array([pos_of[int(v)] for v in ln['from_bus']])
        tb = np.array([pos_of[int(v)] for v in ln['to_bus']])
        lr, lx, lb = _line_parameters(net, base_mva, f_hz, vn[fb])
        on = _flag(ln, 'in_service')
        for k in range(len(ln)):
            add_branch(int(lookup_pos[fb[k]]), int(lookup_pos[tb[k]]), lr[k], lx[k], lb[k], 1.0, 0.0,
                       on[k] and bus_on[fb[k]] and bus_on[tb[k]], 'line', k)

    # ---- two-winding transformers --------------------------------------------------------------
    tr = net['trafo']
    if len(tr):
        hb = np.array([pos_of[int(v)] for v in tr['hv_bus']])
        lb_ = np.array([pos_of[int(v)] for v in tr['lv_bus']])
        side = [s if isinstance(s, str) else '' for s in (tr['tap_side'] if 'tap_side' in tr.columns else [''] * len(tr))]
        vnh, vnl, sh = _tap_voltages(
            _get(tr, 'vn_hv_kv', np.nan), _get(tr, 'vn_lv_kv', np.nan), _get(tr, 'shift_degree', 0.0), side,
            _get(tr, 'tap_pos', np.nan), _get(tr, 'tap_neutral', np.nan), _get(tr, 'tap_step_percent', np.nan),
            _get(tr, 'tap_step_degree', np.nan), _flag(tr, 'tap_phase_shifter', False), calc_angles)
        par = _get(tr, 'parallel', 1.0)
        tr_r, tr_x, tr_b = _trafo_branch(base_mva, _get(tr, 'sn_mva', np.nan), _get(tr, 'vn_lv_kv', np.nan), vnl,
                                         vn[lb_], _get(tr, 'vk_percent', 0.0), _get(tr, 'vkr_percent', 0.0),
                                         _get(tr, 'pfe_kw', 0.0), _get(tr, 'i0_percent', 0.0), par)
        ratio = (vnh / vnl) / (vn[hb] / vn[lb_])          # pandapower `_calc_nominal_ratio_from_dataframe`
        on = _flag(tr, 'in_service')
        for k in range(len(tr)):
            add_branch(int(lookup_pos[hb[k]]), int(lookup_pos[lb_[k]]), tr_r[k], tr_x[k], tr_b[k], ratio[k], sh[k],
                       on[k] and bus_on[hb[k]] and bus_on[lb_[k]], 'trafo', k)

    # ---- three-winding transformers: star equivalent with an auxiliary bus each --------------------
    aux_kv = []
    t3 = net['trafo3w'] if 'trafo3w' in net else None
    if t3 is not None and len(t3):
        eq = _trafo3w_as_two_winding(t3)
        on = _flag(t3, 'in_service')
        for k in range(len(t3)):
            star = nb + len(aux_kv)
            aux_kv.append(float(eq['hv']['vn_hv'][k]))       # the star point sits on the hv voltage level
            ends = {'hv': pos_of[int(t3['hv_bus'].iloc[k])], 'mv': pos_of[int(t3['mv_bus'].iloc[k])],
                    'lv': pos_of[int(t3['lv_bus'].iloc[k])]}
            tside = t3['tap_side'].iloc[k] if 'tap_side' in t3.columns and isinstance(t3['tap_side'].iloc[k], str) else ''
            tpos = _get(t3, 'tap_pos', np.nan)[k]
            tfac = 1.0
            if tside and np.isfinite(tpos) and np.isfinite(_get(t3, 'tap_step_percent', np.nan)[k]):
                tfac = 1.0 + (tpos - _get(t3, 'tap_neutral', 0.0)[k]) * _get(t3, 'tap_step_percent', np.nan)[k] / 100.0
            for side in ('hv', 'mv', 'lv'):
                e = eq[side]
                one = lambda a: np.array([a[k]], dtype=float)
                if side == 'hv':       # winding hv-bus -> star: rated vn_hv / vn_hv
                    v_h, v_l = e['vn_hv'][k] * (tfac if tside == 'hv' else 1.0), e['vn_hv'][k]
                    fb, tb, kv_f, kv_t, rated_l = int(lookup_pos[ends['hv']]), star, vn[ends['hv']], e['vn_hv'][k], e['vn_hv'][k]
                else:                  # star -> mv / lv bus: rated vn_hv / vn_side
                    v_h, v_l = e['vn_hv'][k], e['vn_lv'][k] * (tfac if tside == side else 1.0)
                    fb, tb, kv_f, kv_t, rated_l = star, int(lookup_pos[ends[side]]), e['vn_hv'][k], vn[ends[side]], e['vn_lv'][k]
                rr, xx, bb = _trafo_branch(base_mva, one(e['sn']), np.array([rated_l]), np.array([v_l]), np.array([kv_t]),
                                           one(e['vk']), one(e['vkr']), one(e['pfe']), one(e['i0']), np.ones(1))
                ratio = (v_h / v_l) / (kv_f / kv_t)
                add_branch(fb, tb, rr[0], xx[0], bb[0], ratio, e['shift'][k] if calc_angles else 0.0,
                           on[k] and all(bus_on[v] for v in ends.values()), 'trafo3w', k, side)

    # ---- extended wards (pandapower `_calc_xward_parameter`): an internal bus per xward behind r_ohm + j x_ohm; the voltage
    # source on it is a generator row without active power (below) ------------------------------------------------------------
    xw = net['xward'] if 'xward' in net and len(net['xward']) else None
    xw_aux = {}
    if xw is not None:
        on = _flag(xw, 'in_service')
        for k in range(len(xw)):
            at = pos_of[int(xw['bus'].iloc[k])]
            if not (on[k] and bus_on[at]):
                continue                                 # (pandapower keeps the bus and switches its branch off; nothing hangs on it)
            xw_aux[k] = nb + len(aux_kv)
            aux_kv.append(float(vn[at]))
            base_z = vn[at] ** 2 / base_mva
            add_branch(int(lookup_pos[at]), xw_aux[k], float(xw['r_ohm'].iloc[k]) / base_z, float(xw['x_ohm'].iloc[k]) / base_z,
                       0.0, 1.0, 0.0, True, 'xward', k)

    # ---- impedances (pandapower `_calc_impedance_parameters_from_dataframe`): p.u. on the element's sn_mva -> on the net's;
    # the to-side values go into the asymmetry columns as differences ----------------------------------------------------
    imp = net['impedance'] if 'impedance' in net and len(net['impedance']) else None
    if imp is not None:
        on = _flag(imp, 'in_service')
        to_net = base_mva / _get(imp, 'sn_mva', np.nan)
        rft, xft = _get(imp, 'rft_pu', np.nan) * to_net, _get(imp, 'xft_pu', np.nan) * to_net
        rtf, xtf = _get(imp, 'rtf_pu', np.nan) * to_net, _get(imp, 'xtf_pu', np.nan) * to_net
        for k in range(len(imp)):
            fb, tb = pos_of[int(imp['from_bus'].iloc[k])], pos_of[int(imp['to_bus'].iloc[k])]
            add_branch(int(lookup_pos[fb]), int(lookup_pos[tb]), rft[k], xft[k], 0.0, 1.0, 0.0,
                       on[k] and bus_on[fb] and bus_on[tb], 'impedance', k, r_asym=rtf[k] - rft[k], x_asym=xtf[k] - xft[k])

    # ---- closed bus-bus switches with an impedance (pandapower `_calc_switch_parameter`; runpp's switch_rx_ratio = 2):
    # r = z_ohm * rx / sqrt(1 + rx^2), x = z_ohm / sqrt(1 + rx^2), over the base impedance of the switch's bus ---------------
    if sw is not None:
        rx = 2.0
        for k, (sb, e, et, closed, z) in enumerate(zip(sw['bus'], sw['element'], sw['et'], sw['closed'], sw_z)):
            if et == 'b' and bool(closed) and z > 0.0 and int(sb) in pos_of and int(e) in pos_of:
                fb, tb = pos_of[int(sb)], pos_of[int(e)]
                base_z = vn[fb] ** 2 / base_mva
                add_branch(int(lookup_pos[fb]), int(lookup_pos[tb]), z * rx / np.sqrt(1.0 + rx * rx) / base_z,
                           z / np.sqrt(1.0 + rx * rx) / base_z, 0.0, 1.0, 0.0, bus_on[fb] and bus_on[tb], 'switch', k)

    # ---- open switches at branches (pandapower `_switch_branches`) ------------------------------------
    if sw is not None:
        for et, table in (('l', 'line'), ('t', 'trafo')):
            opened = {}
            for sb, e, k, closed in zip(sw['bus'], sw['element'], sw['et'], sw['closed']):
                if k == et and not bool(closed):
                    opened.setdefault(int(e), set()).add(int(sb))
            for elem, at_buses in opened.items():
                if elem not in net[table].index:
                    continue
                pos = int(net[table].index.get_loc(elem))
                br = next(i for i in range(len(f)) if tbl[i] == table and bpos[i] == pos)
                ends = (int(net[table]['from_bus' if table == 'line' else 'hv_bus'].iloc[pos]),
                        int(net[table]['to_bus' if table == 'line' else 'lv_bus'].iloc[pos]))
                sides = [s for s, eb in enumerate(ends) if eb in at_buses]
                if len(sides) >= 2:
                    status[br] = 0                       # open at both ends: out of service
                elif len(sides) == 1 and status[br]:
                    aux = nb + len(aux_kv)               # open-ended: re-routed to an auxiliary bus
                    aux_kv.append(float(vn[pos_of[ends[sides[0]]]]))
                    if sides[0] == 0:
                        f[br] = aux
                    else:
                        t[br] = aux

    n_all = nb + len(aux_kv)
    base_kv = np.array(base_kv + aux_kv, dtype=float)
    bus_type = np.full(n_all, PQ, dtype=np.int64)
    bus_type[:nb][~np.array(alive_bus, dtype=bool)] = NONE
    pd_, qd_ = np.zeros(n_all), np.zeros(n_all)
    gs, bs = np.zeros(n_all), np.zeros(n_all)
    vm, va = np.ones(n_all), np.zeros(n_all)

    # ---- loads, sgens, storages (pandapower `_calc_pq_elements_and_add_on_ppc`) -------------------------
    for table, sign in (('load', 1.0), ('sgen', -1.0), ('storage', 1.0)):
        df = net[table]
        if not len(df):
            continue
        on = _flag(df, 'in_service')
        sc = _get(df, 'scaling', 1.0)
        pv, qv = df['p_mw'].to_numpy(float), df['q_mvar'].to_numpy(float)     # (a NaN set-point stays NaN: no solution)
        for k, bus in enumerate(df['bus']):
            if on[k]:
                i = bus_lookup[int(bus)]
                pd_[i] += sign * pv[k] * sc[k]
                qd_[i] += sign * qv[k] * sc[k]

    # ---- wards: constant-power part as demand, constant-impedance part as a shunt at 1 p.u.; motors as demand
    # (pandapower `_calc_pq_elements_and_add_on_ppc`, `_get_motor_pq`, `_calc_shunts_and_add_on_ppc`) ---------------------
    for wname in ('ward', 'xward'):
        wd = net[wname] if wname in net and len(net[wname]) else None
        if wd is None:
            continue
        on = _flag(wd, 'in_service')
        for k, bus in enumerate(wd['bus']):
            if on[k]:
                i = bus_lookup[int(bus)]
                pd_[i] += float(wd['ps_mw'].iloc[k]); qd_[i] += float(wd['qs_mvar'].iloc[k])
                gs[i] += float(wd['pz_mw'].iloc[k]); bs[i] -= float(wd['qz_mvar'].iloc[k])
    mt = net['motor'] if 'motor' in net and len(net['motor']) else None
    if mt is not None:
        on = _flag(mt, 'in_service')
        p_m, q_m = motor_pq(mt)
        for k, bus in enumerate(mt['bus']):
            if on[k]:
                pd_[bus_lookup[int(bus)]] += p_m[k]; qd_[bus_lookup[int(bus)]] += q_m[k]

    # ---- shunts (pandapower `_calc_shunts_and_add_on_ppc`) ---------------------------------------------------
    sh_df = net['shunt'] if 'shunt' in net else None
    if sh_df is not None and len(sh_df):
        on = _flag(sh_df, 'in_service')
        step = _get(sh_df, 'step', 1.0)
        for k, bus in enumerate(sh_df['bus']):
            if on[k]:
                i = bus_lookup[int(bus)]
                v_ratio = (base_kv[i] / float(sh_df['vn_kv'].iloc[k])) ** 2
                gs[i] += float(sh_df['p_mw'].iloc[k]) * step[k] * v_ratio
                bs[i] -= float(sh_df['q_mvar'].iloc[k]) * step[k] * v_ratio

    # ---- generators: ext_grids (REF) first, then gens (PV) (pandapower `_build_gen_ppc`) ----------------------------
    g_bus, g_p, g_qmin, g_qmax, g_vg, g_status, g_table, g_pos = [], [], [], [], [], [], [], []
    for k, bus in enumerate(eg['bus'] if len(eg) else []):
        if not eg_on[k]:
            continue
        i = bus_lookup[int(bus)]
        g_bus.append(i); g_p.append(0.0); g_qmin.append(-1e9); g_qmax.append(1e9)
        g_vg.append(float(eg['vm_pu'].iloc[k])); g_status.append(1); g_table.append('ext_grid'); g_pos.append(k)
        bus_type[i] = REF
        vm[i] = float(eg['vm_pu'].iloc[k])
        if calc_angles and 'va_degree' in eg.columns:
            va[i] = float(eg['va_degree'].iloc[k])
    gen = net['gen']
    if len(gen):
        on = _flag(gen, 'in_service')
        sc = _get(gen, 'scaling', 1.0)
        qlo, qhi = _get(gen, 'min_q_mvar', -1e9), _get(gen, 'max_q_mvar', 1e9)
        for k, bus in enumerate(gen['bus']):
            if not on[k]:
                continue
            i = bus_lookup[int(bus)]
            g_bus.append(i); g_p.append(float(gen['p_mw'].iloc[k]) * sc[k]); g_qmin.append(qlo[k]); g_qmax.append(qhi[k])
            g_vg.append(float(gen['vm_pu'].iloc[k])); g_status.append(1); g_table.append('gen'); g_pos.append(k)
            if bus_type[i] != REF:
                bus_type[i] = PV
                vm[i] = float(gen['vm_pu'].iloc[k])

    # ---- DC lines (pandapower `_add_dcline_gens`): two generator rows per line at the end of the generator table — at the to
    # bus what arrives (p_mw less the relative and the fixed losses) at vm_to_pu, at the from bus what is taken, as negative
    # generation, at vm_from_pu; each within its side's reactive range.  g_pos: 2 row for the to end, 2 row + 1 for the from end
    dc = net['dcline'] if 'dcline' in net and len(net['dcline']) else None
    if dc is not None:
        on = _flag(dc, 'in_service')
        for k in range(len(dc)):
            if not on[k]:
                continue
            p_from = float(dc['p_mw'].iloc[k])
            p_to = p_from * (1.0 - _get(dc, 'loss_percent', 0.0)[k] / 100.0) - _get(dc, 'loss_mw', 0.0)[k]
            for end, p_end, side in ((0, p_to, 'to'), (1, -p_from, 'from')):
                i = bus_lookup[int(dc[side + '_bus'].iloc[k])]
                qlo = _get(dc, f'min_q_{side}_mvar', np.nan)[k]
                qhi = _get(dc, f'max_q_{side}_mvar', np.nan)[k]
                g_bus.append(i); g_p.append(p_end); g_qmin.append(-1e9 if np.isnan(qlo) else qlo); g_qmax.append(1e9 if np.isnan(qhi) else qhi)
                g_vg.append(float(dc[f'vm_{side}_pu'].iloc[k])); g_status.append(1); g_table.append('dcline'); g_pos.append(2 * k + end)
                if bus_type[i] != REF:
                    bus_type[i] = PV
                    vm[i] = float(dc[f'vm_{side}_pu'].iloc[k])

    for k, aux in xw_aux.items():                    # (after the generators, as pandapower orders its gen table)
        g_bus.append(aux); g_p.append(0.0); g_qmin.append(-1e9); g_qmax.append(1e9)
        g_vg.append(float(xw['vm_pu'].iloc[k])); g_status.append(1); g_table.append('xward'); g_pos.append(k)
        bus_type[aux] = PV
        vm[aux] = float(xw['vm_pu'].iloc[k])

    ppc = PPC(base_mva=base_mva, bus_type=bus_type, pd=pd_, qd=qd_, gs=gs, bs=bs, vm=vm, va=va, base_kv=base_kv,
              f=np.array(f, dtype=np.int64), t=np.array(t, dtype=np.int64), r=np.array(r, dtype=float),
              x=np.array(x, dtype=float), b=np.array(b, dtype=complex), tap=np.array(tap, dtype=float),
              shift=np.array(shift, dtype=float), status=np.array(status, dtype=np.int64), br_table=tbl,
              br_pos=np.array(bpos, dtype=np.int64), br_side=bside,
              g_bus=np.array(g_bus, dtype=np.int64), g_p=np.array(g_p, dtype=float),
              g_qmin=np.array(g_qmin, dtype=float), g_qmax=np.array(g_qmax, dtype=float),
              g_vg=np.array(g_vg, dtype=float), g_status=np.array(g_status, dtype=np.int64), g_table=g_table,
              g_pos=np.array(g_pos, dtype=np.int64), bus_lookup=bus_lookup, n_net_buses=nb, calc_angles=calc_angles,
              r_asym=np.array(r_as, dtype=float), x_asym=np.array(x_as, dtype=float))
    if not (ppc.bus_type == REF).any():
        raise ValueError('net has no in-service ext_grid (no slack bus)')
    if check_connectivity:
        isolate_unsupplied(ppc)
    return ppc


def supplied_buses(ppc: PPC, status=None) -> np.ndarray:
    """Buses with a path of in-service branches to a REF bus."""
    st = ppc.status if status is None else np.asarray(status)
    adj = [[] for _ in range(ppc.nb)]
    for k in range(ppc.nbr):
        if st[k]:
            adj[int(ppc.f[k])].append(int(ppc.t[k]))
            adj[int(ppc.t[k])].append(int(ppc.f[k]))
    seen = np.zeros(ppc.nb, dtype=bool)
    stack = [int(i) for i in np.flatnonzero(ppc.bus_type == REF)]
    for i in stack:
        seen[i] = True
    while stack:
        u = stack.pop()
        for w in adj[u]:
            if not seen[w] and ppc.bus_type[w] != NONE:
                seen[w] = True
                stack.append(w)
    return seen


def isolate_unsupplied(ppc: PPC) -> None:
    """pandapower `check_connectivity=True`: buses that no slack supplies are taken out (type NONE)
    together with their generators; their branches are switched off."""
    ok = supplied_buses(ppc)
    ppc.bus_type[~ok] = NONE
    for k in range(ppc.nbr):
        if not (ok[ppc.f[k]] and ok[ppc.t[k]]):
            ppc.status[k] = 0
    for g in range(len(ppc.g_bus)):
        if not ok[ppc.g_bus[g]]:
            ppc.g_status[g] = 0


def ppc_from_matrices(base_mva, bus, branch, gen) -> PPC:
    """A case given directly as pypower `bus` / `branch` / `gen` matrices with consecutive 0-based
    bus numbers (the published IEEE / textbook cases of the known-answer tests)."""
    bus, branch, gen = (np.asarray(a, dtype=float) for a in (bus, branch, gen))
    nb = bus.shape[0]
    g_on = gen[:, 7] > 0
    ppc = PPC(base_mva=float(base_mva), bus_type=bus[:, 1].astype(np.int64), pd=bus[:, 2].copy(), qd=bus[:, 3].copy(),
              gs=bus[:, 4].copy(), bs=bus[:, 5].copy(), vm=bus[:, 7].copy(), va=bus[:, 8].copy(),
              base_kv=bus[:, 9].copy(), f=branch[:, 0].astype(np.int64), t=branch[:, 1].astype(np.int64),
              r=branch[:, 2].copy(), x=branch[:, 3].copy(), b=branch[:, 4].astype(complex),
              tap=np.where(branch[:, 8] == 0, 1.0, branch[:, 8]), shift=branch[:, 9].copy(),
              status=(branch[:, 10] > 0).astype(np.int64), br_table=['line'] * len(branch),
              br_pos=np.arange(len(branch)), br_side=[''] * len(branch),
              g_bus=gen[:, 0].astype(np.int64), g_p=gen[:, 1].copy(), g_qmin=gen[:, 4].copy(), g_qmax=gen[:, 3].copy(),
              g_vg=gen[:, 5].copy(), g_status=g_on.astype(np.int64),
              g_table=['ext_grid' if bus[int(g[0]), 1] == REF else 'gen' for g in gen], g_pos=np.arange(len(gen)),
              bus_lookup={i: i for i in range(nb)}, n_net_buses=nb, calc_angles=True)
    for g in np.flatnonzero(g_on):                       # generator set-points define |V| at PV / REF buses
        i = int(ppc.g_bus[g])
        if ppc.bus_type[i] in (PV, REF):
            ppc.vm[i] = ppc.g_vg[g]
    ppc.vm[ppc.bus_type == PQ] = 1.0
    return ppc
