"""MATPOWER/pypower-format case I/O.

`case_from_ppc` builds the per-unit :class:`opfgym_amd.case.Case` that crosses
the C ABI directly from pypower-style `bus` / `branch` / `gen` matrices — the
form pandapower itself hands to its solver (`net._ppc['internal']`, third
party).  On a machine that has pandapower, `scripts/export_pandapower_case.py`
dumps those matrices (plus pandapower's own results) to an .npz; loading that
file here by-passes this repository's restatement of the element models
(`case.net_to_case`) and makes a 1e-6 comparison against pandapower possible.
"""
from __future__ import annotations

import numpy as np

from .case import PQ, PV, REF, Case, _propagate_angles, branch_stamps

# pypower column indices (idx_bus / idx_brch / idx_gen)
BUS_I, BUS_TYPE, PD, QD, GS, BS, VM, VA, BASE_KV = 0, 1, 2, 3, 4, 5, 7, 8, 9
F_BUS, T_BUS, BR_R, BR_X, BR_B, RATE_A, TAP, SHIFT, BR_STATUS = 0, 1, 2, 3, 4, 5, 8, 9, 10
GEN_BUS, PG, QG, QMAX, QMIN, VG, GEN_STATUS = 0, 1, 2, 3, 4, 5, 7


def case_from_ppc(base_mva, bus, branch, gen, br_g=None):
    """bus/branch/gen: pypower matrices with consecutive 0-based bus numbers
    (`ppci`).  Returns (case, p_inj, q_inj, qg_min, qg_max): the injections are
    generation − demand per bus in p.u. (gen Q excluded), limits in p.u."""
    bus, branch, gen = (np.asarray(a, dtype=float) for a in (bus, branch, gen))
    nb = bus.shape[0]
    on = branch[:, BR_STATUS] > 0
    br = branch[on]
    bc = br[:, BR_B].astype(complex)
    if br_g is not None:
        bc = bc - 1j * np.asarray(br_g, dtype=float)[on]       # j*bc/2 = (g + jb)/2 per side
    ratio = np.where(br[:, TAP] == 0, 1.0, br[:, TAP])
    shift = np.deg2rad(br[:, SHIFT])
    yff, yft, ytf, ytt = branch_stamps(br[:, BR_R], br[:, BR_X], bc, ratio, shift)
    bus_type = bus[:, BUS_TYPE].astype(np.int32)
    if not np.isin(bus_type, (PQ, PV, REF)).all():
        raise ValueError('isolated buses (type 4) must be removed first (ppci has none)')
    vm_set = np.where(bus_type == PQ, 1.0, bus[:, VM])
    g_on = gen[:, GEN_STATUS] > 0
    p = -bus[:, PD] / base_mva
    q = -bus[:, QD] / base_mva
    qmin = np.full(nb, -np.inf)
    qmax = np.full(nb, np.inf)
    acc_lo, acc_hi, has = np.zeros(nb), np.zeros(nb), np.zeros(nb, bool)
    for row in gen[g_on]:
        i = int(row[GEN_BUS])
        p[i] += row[PG] / base_mva
        vm_set[i] = row[VG] if bus_type[i] != PQ else vm_set[i]
        acc_lo[i] += row[QMIN] / base_mva
        acc_hi[i] += row[QMAX] / base_mva
        has[i] = True
    qmin[has], qmax[has] = acc_lo[has], acc_hi[has]
    f, t = br[:, F_BUS].astype(np.int32), br[:, T_BUS].astype(np.int32)
    va_ref = np.where(bus_type == REF, np.deg2rad(bus[:, VA]), 0.0)
    va0 = _propagate_angles(nb, f, t, shift, bus_type, va_ref)
    rate = np.where(br[:, RATE_A] > 0, br[:, RATE_A], np.inf)
    case = Case(
        base_mva=float(base_mva), bus_type=bus_type, vn_kv=bus[:, BASE_KV].copy(), vm_set=vm_set,
        va_set=np.where(bus_type == REF, va_ref, va0), gs=bus[:, GS] / base_mva, bs=bus[:, BS] / base_mva,
        f=f, t=t, yff=yff, yft=yft, ytf=ytf, ytt=ytt,
        # percent of RATE_A [MVA] at nominal voltage: |I| p.u. * baseMVA / rate * 100
        kf=base_mva / rate * 100.0, kt=base_mva / rate * 100.0,
        br_kind=np.zeros(len(f), dtype=np.int32), br_elem=np.flatnonzero(on).astype(np.int32),
        br_side=np.zeros(len(f), dtype=np.int32),
        bus_lookup={i: i for i in range(nb)}, ref_elems=np.zeros(0, dtype=np.int32),
        meta={'source': 'ppc'})
    return case, p, q, qmin, qmax


def load_exported_case(path):
    """Read an .npz written by scripts/export_pandapower_case.py."""
    z = np.load(path, allow_pickle=False)
    case, p, q, qmin, qmax = case_from_ppc(float(z['baseMVA']), z['bus'], z['branch'], z['gen'],
                                           z['br_g'] if 'br_g' in z else None)
    ref = {k[4:]: z[k] for k in z.files if k.startswith('res_')}
    return case, p, q, qmin, qmax, ref
