"""Minimal pandapower-shaped network container.

The reference environment (`/root/reference/opfgym/opf_env.py:58`) mutates a
``pandapowerNet``: an attribute-dict of pandas DataFrames (``net.load``,
``net['load']``, ``net.res_bus`` ...).  pandapower itself is a third-party
dependency that is not available here, so this module provides the same
*shape* (table names and column names) with a handful of ``create_*`` helpers.
A real pandapowerNet can be passed wherever a :class:`Net` is accepted: only
the tables/columns listed in SURVEY.md Appendix B are touched.
"""
from __future__ import annotations

import copy

import numpy as np
import pandas as pd

_TABLES = {
    'bus': ['name', 'vn_kv', 'type', 'in_service', 'min_vm_pu', 'max_vm_pu'],
    'line': ['name', 'from_bus', 'to_bus', 'length_km', 'r_ohm_per_km',
             'x_ohm_per_km', 'c_nf_per_km', 'g_us_per_km', 'max_i_ka', 'df',
             'parallel', 'in_service', 'max_loading_percent'],
    'trafo': ['name', 'hv_bus', 'lv_bus', 'sn_mva', 'vn_hv_kv', 'vn_lv_kv',
              'vk_percent', 'vkr_percent', 'pfe_kw', 'i0_percent',
              'shift_degree', 'tap_side', 'tap_neutral', 'tap_pos',
              'tap_step_percent', 'parallel', 'df', 'in_service',
              'max_loading_percent'],
    'trafo3w': ['name', 'hv_bus', 'mv_bus', 'lv_bus', 'sn_hv_mva', 'sn_mv_mva', 'sn_lv_mva', 'vn_hv_kv',
                'vn_mv_kv', 'vn_lv_kv', 'vk_hv_percent', 'vk_mv_percent', 'vk_lv_percent', 'vkr_hv_percent',
                'vkr_mv_percent', 'vkr_lv_percent', 'pfe_kw', 'i0_percent', 'shift_mv_degree', 'shift_lv_degree',
                'tap_side', 'tap_neutral', 'tap_pos', 'tap_step_percent', 'in_service', 'max_loading_percent'],
    'load': ['name', 'bus', 'p_mw', 'q_mvar', 'scaling', 'in_service',
             'controllable'],
    'sgen': ['name', 'bus', 'p_mw', 'q_mvar', 'scaling', 'in_service',
             'controllable'],
    'storage': ['name', 'bus', 'p_mw', 'q_mvar', 'scaling', 'in_service',
                'controllable', 'min_p_mw', 'max_p_mw', 'min_q_mvar', 'max_q_mvar'],
    'gen': ['name', 'bus', 'p_mw', 'vm_pu', 'scaling', 'in_service',
            'controllable', 'min_q_mvar', 'max_q_mvar'],
    'ext_grid': ['name', 'bus', 'vm_pu', 'va_degree', 'in_service'],
    'shunt': ['bus', 'p_mw', 'q_mvar', 'vn_kv', 'step', 'in_service'],
    'switch': ['bus', 'element', 'et', 'closed'],
    # pandapower element types beyond those of the SimBench grids (static in the batched environment: no actuator / sampling)
    'ward': ['name', 'bus', 'ps_mw', 'qs_mvar', 'pz_mw', 'qz_mvar', 'in_service'],
    'xward': ['name', 'bus', 'ps_mw', 'qs_mvar', 'pz_mw', 'qz_mvar', 'r_ohm', 'x_ohm', 'vm_pu', 'in_service'],
    'dcline': ['name', 'from_bus', 'to_bus', 'p_mw', 'loss_percent', 'loss_mw', 'vm_from_pu', 'vm_to_pu', 'max_p_mw',
               'min_q_from_mvar', 'min_q_to_mvar', 'max_q_from_mvar', 'max_q_to_mvar', 'in_service'],
    'impedance': ['name', 'from_bus', 'to_bus', 'rft_pu', 'xft_pu', 'rtf_pu', 'xtf_pu', 'sn_mva', 'in_service'],
    'motor': ['name', 'bus', 'pn_mech_mw', 'cos_phi', 'efficiency_percent', 'loading_percent', 'scaling', 'in_service'],
    'poly_cost': ['element', 'et', 'cp0_eur', 'cp1_eur_per_mw',
                  'cp2_eur_per_mw2', 'cq0_eur', 'cq1_eur_per_mvar',
                  'cq2_eur_per_mvar2'],
    'pwl_cost': ['power_type', 'element', 'et', 'points'],
}


_OBJ_COLS = ('name', 'type', 'tap_side', 'et', 'points', 'power_type')
_BOOL_COLS = ('in_service', 'controllable', 'closed')
_INT_COLS = ('bus', 'from_bus', 'to_bus', 'hv_bus', 'mv_bus', 'lv_bus', 'element', 'parallel')


def _dtype_of(col):
    if col in _OBJ_COLS:
        return object
    if col in _BOOL_COLS:
        return bool
    if col in _INT_COLS:
        return np.int64
    return np.float64


class Net(dict):
    """Attribute-dict of DataFrames, same access idioms as a pandapowerNet
    (`net.load` and `net['load']`, both used by opf_env.py:105,267)."""

    def __init__(self, name: str = '', f_hz: float = 50.0, sn_mva: float = 1.0):
        super().__init__()
        self['name'] = name
        self['f_hz'] = float(f_hz)
        self['sn_mva'] = float(sn_mva)
        for tbl, cols in _TABLES.items():
            self[tbl] = pd.DataFrame({c: pd.Series(dtype=_dtype_of(c)) for c in cols})

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError as exc:
            raise AttributeError(key) from exc

    def __setattr__(self, key, value):
        self[key] = value

    def __deepcopy__(self, memo):
        new = Net.__new__(Net)
        dict.__init__(new)
        for k, v in self.items():
            new[k] = copy.deepcopy(v, memo)
        return new

    def deepcopy(self) -> 'Net':
        return copy.deepcopy(self)


def _append(net, table: str, row: dict, index=None) -> int:
    df = net[table]
    if index is None:
        index = 0 if len(df) == 0 else int(df.index.max()) + 1
    # rows are collected as objects (`finalize` gives the columns their dtypes).  The frame is rebuilt from one object
    # array per column — existing values copied by numpy, the new one appended — rather than enlarged through `.loc`:
    # pandas' enlargement concatenates and guesses a dtype for all-missing entries (deprecated), and a cell-by-cell
    # rebuild in Python made 2 000 `create_bus` calls take a minute
    keep = np.asarray(df.index != index) if len(df) else np.zeros(0, bool)          # (an existing index is overwritten)
    cols = list(df.columns) + [c for c in row if c not in df.columns]
    data = {}
    for c in cols:
        old = df[c].to_numpy(dtype=object)[keep] if c in df.columns else np.full(int(keep.sum()), np.nan, dtype=object)
        col = np.empty(len(old) + 1, dtype=object)
        col[:-1] = old
        col[-1] = row.get(c, np.nan)
        data[c] = col
    net[table] = pd.DataFrame(data, index=list(df.index[keep]) + [index], columns=cols)
    return index


def _finalize_dtypes(net) -> None:
    """Give numeric columns numeric dtypes (rows are appended as objects)."""
    for tbl in _TABLES:
        df = net[tbl]
        if len(df) == 0:
            continue
        for col in df.columns:
            if col in ('name', 'type', 'tap_side', 'et', 'points', 'power_type'):
                continue
            if col in ('in_service', 'controllable', 'closed'):
                df[col] = df[col].astype(bool)
                continue
            try:
                conv = pd.to_numeric(df[col])
            except (ValueError, TypeError):
                continue
            if col in ('bus', 'from_bus', 'to_bus', 'hv_bus', 'mv_bus', 'lv_bus',
                       'element', 'parallel') and not conv.isna().any():
                conv = conv.astype(np.int64)
            df[col] = conv


def create_bus(net, vn_kv, name=None, index=None, in_service=True, **kw) -> int:
    return _append(net, 'bus', dict(name=name, vn_kv=float(vn_kv), type='b',
                                    in_service=bool(in_service), **kw), index)


def create_line_from_parameters(net, from_bus, to_bus, length_km, r_ohm_per_km,
                                x_ohm_per_km, c_nf_per_km, max_i_ka,
                                g_us_per_km=0.0, df=1.0, parallel=1,
                                in_service=True, name=None, index=None, **kw) -> int:
    return _append(net, 'line', dict(
        name=name, from_bus=int(from_bus), to_bus=int(to_bus),
        length_km=float(length_km), r_ohm_per_km=float(r_ohm_per_km),
        x_ohm_per_km=float(x_ohm_per_km), c_nf_per_km=float(c_nf_per_km),
        g_us_per_km=float(g_us_per_km), max_i_ka=float(max_i_ka), df=float(df),
        parallel=int(parallel), in_service=bool(in_service), **kw), index)


def create_transformer_from_parameters(net, hv_bus, lv_bus, sn_mva, vn_hv_kv,
                                       vn_lv_kv, vk_percent, vkr_percent,
                                       pfe_kw, i0_percent, shift_degree=0.0,
                                       tap_side=None, tap_neutral=0, tap_pos=0,
                                       tap_step_percent=0.0, parallel=1, df=1.0,
                                       in_service=True, name=None, index=None,
                                       **kw) -> int:
    return _append(net, 'trafo', dict(
        name=name, hv_bus=int(hv_bus), lv_bus=int(lv_bus), sn_mva=float(sn_mva),
        vn_hv_kv=float(vn_hv_kv), vn_lv_kv=float(vn_lv_kv),
        vk_percent=float(vk_percent), vkr_percent=float(vkr_percent),
        pfe_kw=float(pfe_kw), i0_percent=float(i0_percent),
        shift_degree=float(shift_degree), tap_side=tap_side,
        tap_neutral=float(tap_neutral), tap_pos=float(tap_pos),
        tap_step_percent=float(tap_step_percent), parallel=int(parallel),
        df=float(df), in_service=bool(in_service), **kw), index)


def create_transformer3w_from_parameters(net, hv_bus, mv_bus, lv_bus, vn_hv_kv, vn_mv_kv, vn_lv_kv, sn_hv_mva,
                                         sn_mv_mva, sn_lv_mva, vk_hv_percent, vk_mv_percent, vk_lv_percent,
                                         vkr_hv_percent, vkr_mv_percent, vkr_lv_percent, pfe_kw, i0_percent,
                                         shift_mv_degree=0.0, shift_lv_degree=0.0, tap_side=None, tap_neutral=0,
                                         tap_pos=0, tap_step_percent=np.nan, in_service=True, name=None, index=None,
                                         **kw) -> int:
    """Same parameters as pandapower.create_transformer3w_from_parameters (vk_hv: hv-mv, vk_mv: mv-lv,
    vk_lv: hv-lv short-circuit voltage)."""
    return _append(net, 'trafo3w', dict(
        name=name, hv_bus=int(hv_bus), mv_bus=int(mv_bus), lv_bus=int(lv_bus), sn_hv_mva=float(sn_hv_mva),
        sn_mv_mva=float(sn_mv_mva), sn_lv_mva=float(sn_lv_mva), vn_hv_kv=float(vn_hv_kv), vn_mv_kv=float(vn_mv_kv),
        vn_lv_kv=float(vn_lv_kv), vk_hv_percent=float(vk_hv_percent), vk_mv_percent=float(vk_mv_percent),
        vk_lv_percent=float(vk_lv_percent), vkr_hv_percent=float(vkr_hv_percent), vkr_mv_percent=float(vkr_mv_percent),
        vkr_lv_percent=float(vkr_lv_percent), pfe_kw=float(pfe_kw), i0_percent=float(i0_percent),
        shift_mv_degree=float(shift_mv_degree), shift_lv_degree=float(shift_lv_degree), tap_side=tap_side,
        tap_neutral=float(tap_neutral), tap_pos=float(tap_pos), tap_step_percent=float(tap_step_percent),
        in_service=bool(in_service), **kw), index)


def _create_unit(net, table, bus, p_mw, q_mvar, scaling, in_service, name,
                 index, controllable, **kw) -> int:
    return _append(net, table, dict(
        name=name, bus=int(bus), p_mw=float(p_mw), q_mvar=float(q_mvar),
        scaling=float(scaling), in_service=bool(in_service),
        controllable=bool(controllable), **kw), index)


def create_load(net, bus, p_mw, q_mvar=0.0, scaling=1.0, in_service=True,
                name=None, index=None, controllable=False, **kw) -> int:
    return _create_unit(net, 'load', bus, p_mw, q_mvar, scaling, in_service,
                        name, index, controllable, **kw)


def create_sgen(net, bus, p_mw, q_mvar=0.0, scaling=1.0, in_service=True,
                name=None, index=None, controllable=False, **kw) -> int:
    return _create_unit(net, 'sgen', bus, p_mw, q_mvar, scaling, in_service,
                        name, index, controllable, **kw)


def create_storage(net, bus, p_mw, q_mvar=0.0, scaling=1.0, in_service=True,
                   name=None, index=None, controllable=False, **kw) -> int:
    return _create_unit(net, 'storage', bus, p_mw, q_mvar, scaling, in_service,
                        name, index, controllable, **kw)


def create_gen(net, bus, p_mw, vm_pu=1.0, scaling=1.0, in_service=True,
               min_q_mvar=np.nan, max_q_mvar=np.nan, name=None, index=None,
               controllable=True, **kw) -> int:
    return _append(net, 'gen', dict(
        name=name, bus=int(bus), p_mw=float(p_mw), vm_pu=float(vm_pu),
        scaling=float(scaling), in_service=bool(in_service),
        controllable=bool(controllable), min_q_mvar=float(min_q_mvar),
        max_q_mvar=float(max_q_mvar), **kw), index)


def create_ext_grid(net, bus, vm_pu=1.0, va_degree=0.0, in_service=True,
                    name=None, index=None, **kw) -> int:
    return _append(net, 'ext_grid', dict(
        name=name, bus=int(bus), vm_pu=float(vm_pu), va_degree=float(va_degree),
        in_service=bool(in_service), **kw), index)


def create_shunt(net, bus, q_mvar, p_mw=0.0, vn_kv=None, step=1,
                 in_service=True, index=None) -> int:
    if vn_kv is None:
        vn_kv = float(net.bus.at[bus, 'vn_kv'])
    return _append(net, 'shunt', dict(bus=int(bus), p_mw=float(p_mw),
                                      q_mvar=float(q_mvar), vn_kv=float(vn_kv),
                                      step=int(step),
                                      in_service=bool(in_service)), index)


def create_switch(net, bus, element, et, closed=True, index=None, z_ohm=0.0) -> int:
    row = dict(bus=int(bus), element=int(element), et=et, closed=bool(closed))
    if z_ohm or 'z_ohm' in net['switch'].columns:          # (the column exists only in nets that use it)
        row['z_ohm'] = float(z_ohm)
    return _append(net, 'switch', row, index)


def create_ward(net, bus, ps_mw, qs_mvar, pz_mw, qz_mvar, in_service=True, name=None, index=None) -> int:
    """pandapower.create_ward: a constant-power part (ps, qs) and a constant-impedance part (pz, qz at 1 p.u.), consumption
    positive."""
    return _append(net, 'ward', dict(name=name, bus=int(bus), ps_mw=float(ps_mw), qs_mvar=float(qs_mvar), pz_mw=float(pz_mw),
                                     qz_mvar=float(qz_mvar), in_service=bool(in_service)), index)


def create_xward(net, bus, ps_mw, qs_mvar, pz_mw, qz_mvar, r_ohm, x_ohm, vm_pu, in_service=True, name=None, index=None) -> int:
    """pandapower.create_xward: a ward plus, behind r_ohm + j x_ohm, an internal voltage source at vm_pu (no active power)."""
    return _append(net, 'xward', dict(name=name, bus=int(bus), ps_mw=float(ps_mw), qs_mvar=float(qs_mvar), pz_mw=float(pz_mw),
                                      qz_mvar=float(qz_mvar), r_ohm=float(r_ohm), x_ohm=float(x_ohm), vm_pu=float(vm_pu),
                                      in_service=bool(in_service)), index)


def create_dcline(net, from_bus, to_bus, p_mw, loss_percent, loss_mw, vm_from_pu, vm_to_pu, max_p_mw=np.nan,
                  min_q_from_mvar=np.nan, min_q_to_mvar=np.nan, max_q_from_mvar=np.nan, max_q_to_mvar=np.nan, in_service=True,
                  name=None, index=None) -> int:
    """pandapower.create_dcline: p_mw leaves the grid at from_bus, p_mw (1 - loss_percent / 100) - loss_mw enters it at to_bus;
    both ends hold their bus voltage within their reactive range."""
    return _append(net, 'dcline', dict(
        name=name, from_bus=int(from_bus), to_bus=int(to_bus), p_mw=float(p_mw), loss_percent=float(loss_percent),
        loss_mw=float(loss_mw), vm_from_pu=float(vm_from_pu), vm_to_pu=float(vm_to_pu), max_p_mw=float(max_p_mw),
        min_q_from_mvar=float(min_q_from_mvar), min_q_to_mvar=float(min_q_to_mvar), max_q_from_mvar=float(max_q_from_mvar),
        max_q_to_mvar=float(max_q_to_mvar), in_service=bool(in_service)), index)


def create_impedance(net, from_bus, to_bus, rft_pu, xft_pu, sn_mva, rtf_pu=None, xtf_pu=None, in_service=True, name=None,
                     index=None) -> int:
    """pandapower.create_impedance: a series impedance in per unit of its own `sn_mva`, which may differ by direction."""
    return _append(net, 'impedance', dict(
        name=name, from_bus=int(from_bus), to_bus=int(to_bus), rft_pu=float(rft_pu), xft_pu=float(xft_pu),
        rtf_pu=float(rft_pu if rtf_pu is None else rtf_pu), xtf_pu=float(xft_pu if xtf_pu is None else xtf_pu),
        sn_mva=float(sn_mva), in_service=bool(in_service)), index)


def create_motor(net, bus, pn_mech_mw, cos_phi, efficiency_percent=100.0, loading_percent=100.0, scaling=1.0,
                 in_service=True, name=None, index=None) -> int:
    """pandapower.create_motor (the columns the power flow reads)."""
    return _append(net, 'motor', dict(name=name, bus=int(bus), pn_mech_mw=float(pn_mech_mw), cos_phi=float(cos_phi),
                                      efficiency_percent=float(efficiency_percent), loading_percent=float(loading_percent),
                                      scaling=float(scaling), in_service=bool(in_service)), index)


def create_poly_cost(net, element, et, cp1_eur_per_mw, cp0_eur=0.0,
                     cq1_eur_per_mvar=0.0, cq0_eur=0.0, cp2_eur_per_mw2=0.0,
                     cq2_eur_per_mvar2=0.0, index=None) -> int:
    """Same argument order as pandapower.create_poly_cost as it is called at
    voltage_control.py:88-100, eco_dispatch.py:97-99, max_renewable.py:96."""
    return _append(net, 'poly_cost', dict(
        element=int(element), et=et, cp0_eur=float(cp0_eur),
        cp1_eur_per_mw=float(cp1_eur_per_mw),
        cp2_eur_per_mw2=float(cp2_eur_per_mw2), cq0_eur=float(cq0_eur),
        cq1_eur_per_mvar=float(cq1_eur_per_mvar),
        cq2_eur_per_mvar2=float(cq2_eur_per_mvar2)), index)


def create_pwl_cost(net, element, et, points, power_type='p', index=None) -> int:
    """Same shape as pandapower.create_pwl_cost (eco_dispatch.py:95)."""
    return _append(net, 'pwl_cost', dict(
        power_type=power_type, element=int(element), et=et,
        points=[list(map(float, p)) for p in points]), index)


def finalize(net) -> Net:
    _finalize_dtypes(net)
    return net
