"""Drop-in for the reference's `power_flow_solver` seam (batch = 1 plumbing).

`OpfEnv(..., power_flow_solver=opfgym_amd.power_flow_solver)` replaces the
`pp.runpp(net, enforce_q_lims=True)` call of `OpfEnv.default_power_flow`
(/root/reference/opfgym/opf_env.py:53,70,657,696-709; also reached from
security_constrained.py:53 and reward.py:190).  Same contract: mutate the
`net.res_*` tables in place, raise `LoadflowNotConverged` on failure.

The solve itself is `opfx_solve` with B = 1 on the GPU (no CPU fallback); only
the table <-> per-unit conversion and the writing of result columns happen on
the host, as they do inside pandapower.
"""
from __future__ import annotations

import numpy as np
import pandas as pd

from . import capi
from .case import (KIND_IMPEDANCE, KIND_XWARD, KIND_TRAFO3W, KIND_LINE, KIND_TRAFO, REF, _table, bus_injections, expand_dclines,
                   generator_dispatch, net_to_case, static_consumption)


class LoadflowNotConverged(Exception):
    """Raised when the Newton-Raphson solve does not converge.  When pandapower is
    importable its own `pp.powerflow.LoadflowNotConverged` is raised instead so
    that `OpfEnv.run_power_flow` (opf_env.py:660) catches it."""


def _not_converged_exception():
    try:
        import pandapower as pp
        return pp.powerflow.LoadflowNotConverged
    except Exception:
        return LoadflowNotConverged


class BatchedPowerFlowSolver:
    def __init__(self, device='cuda:0', tolerance=1e-8, max_iteration=10, debug=None):
        self.device = device
        self.debug = capi.debug_opts(debug)          # developer switches (include/opfx_debug.h); never from the environment
        self.tol, self.max_it = tolerance, max_iteration
        self._cache = {}
        self.max_cached_plans = 256

    def _context(self, case):
        key = (case.nb, case.nbr, case.bus_type.tobytes(), case.f.tobytes(), case.t.tobytes(),
               case.yff.tobytes(), case.yft.tobytes(), case.ytf.tobytes(), case.ytt.tobytes(),
               case.vm_set.tobytes(), case.va_set.tobytes(), case.gs.tobytes(), case.bs.tobytes(),
               # (the loading factors and the DC model are part of the compiled plan too: two nets with the same admittances
               #  and other ratings — max_i_ka, sn_mva, df, a rated voltage — must not share one; found by tests/metamorphic.py)
               case.kf.tobytes(), case.kt.tobytes(), float(case.base_mva),
               None if case.bdc is None else case.bdc.tobytes(), None if case.pfinj is None else case.pfinj.tobytes())
        # Compiled plans are kept per topology (most recently used last): the reference's N-1 loop
        # (security_constrained.py:44-62) toggles one element at a time and comes back to every topology
        # in every step, so each contingency is compiled once, not once per call.
        ctx = self._cache.pop(key, None)
        if ctx is None:
            import torch
            dev = torch.device(self.device)
            ctx = capi.Context(capi.Plan(case, debug=self.debug), dev.index or 0, debug=self.debug)
            while len(self._cache) >= self.max_cached_plans:
                self._cache.pop(next(iter(self._cache)))
        self._cache[key] = ctx
        return ctx

    def __call__(self, net, enforce_q_lims=True, **kwargs):
        """Keyword arguments as `pp.runpp` where they apply: `calculate_voltage_angles`, `init` ('flat' — the default
        here —, 'dc', or pandapower's own default 'auto': 'dc' when voltage angles are calculated)."""
        import torch
        case = net_to_case(net, kwargs.get('calculate_voltage_angles', 'auto'))
        init = kwargs.get('init', 'flat')
        if init not in ('flat', 'dc', 'auto'):
            raise NotImplementedError(f"init={init!r}: 'flat', 'dc' or 'auto' (no 'results' start)")
        ctx = self._context(case)
        if init == 'auto':       # (no DC model — a zero-reactance branch has no finite 1/x —: flat, as for grids below 70 kV)
            init = 'dc' if case.meta.get('calc_angles') and ctx.plan.info['has_dc'] else 'flat'
        base = case.base_mva
        p, q, qmin, qmax = bus_injections(net, case)
        dev = torch.device(self.device)
        as_t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=dev)
        out = capi.solve(ctx, as_t(p[None] / base), as_t(q[None] / base), qg_min=as_t(qmin / base),
                         qg_max=as_t(qmax / base), tol=self.tol, max_iter=self.max_it, init=init,
                         enforce_q_lims=bool(enforce_q_lims) and case.bus_type.tolist().count(2) > 0)
        self.last_iterations = int(out['iterations'][0])
        if not bool(out['converged'][0]):
            raise _not_converged_exception()('batched Newton-Raphson did not converge')
        self._write_results(net, case, {k: v[0].cpu().numpy() for k, v in out.items()}, p, q)

    @staticmethod
    def _write_results(net, case, r, p_mw, q_mvar):
        base = case.base_mva
        nbus = len(net['bus'])
        vm, va = np.full(nbus, np.nan), np.full(nbus, np.nan)
        for pos, b in enumerate(net['bus'].index):
            if int(b) in case.bus_lookup:
                i = case.bus_lookup[int(b)]
                vm[pos], va[pos] = r['vm'][i], np.degrees(r['va'][i])
        net['res_bus'] = pd.DataFrame({'vm_pu': vm, 'va_degree': va}, index=net['bus'].index)
        for tbl, kind in (('line', KIND_LINE), ('trafo', KIND_TRAFO)):
            # part of the solved case: its loading (a branch behind ONE open switch is, it still carries its
            # charging current).  Not part of it (out of service, open at both ends): 0 MVA over the voltages
            # of its end buses, as pandapower computes i_ka — 0 % between energised buses, NaN next to a
            # de-energised one.
            load = np.full(len(net[tbl]), np.nan)
            sel = case.br_kind == kind
            load[case.br_elem[sel]] = r['loading'][sel]
            ends = ('from_bus', 'to_bus') if tbl == 'line' else ('hv_bus', 'lv_bus')
            in_case = np.zeros(len(net[tbl]), bool)
            in_case[case.br_elem[sel]] = True
            for pos in np.flatnonzero(~in_case):
                alive = all(int(net[tbl][e].iloc[pos]) in case.bus_lookup and
                            not np.isnan(vm[net['bus'].index.get_loc(int(net[tbl][e].iloc[pos]))]) for e in ends)
                load[pos] = 0.0 if alive else np.nan
            net['res_' + tbl] = pd.DataFrame({'loading_percent': load}, index=net[tbl].index)
        if 'trafo3w' in net and len(net['trafo3w']):
            # the worst of the three windings (NaN-propagating, as numpy's max); out of service: 0 %
            ld3 = np.zeros(len(net['trafo3w']))
            for pos in range(len(ld3)):
                sel = (case.br_kind == KIND_TRAFO3W) & (case.br_elem == pos)
                if sel.any():
                    ld3[pos] = np.max(r['loading'][sel])
            net['res_trafo3w'] = pd.DataFrame({'loading_percent': ld3}, index=net['trafo3w'].index)
        ref_buses = np.flatnonzero(case.bus_type == REF)
        ordinal = {int(b): k for k, b in enumerate(ref_buses)}
        # generation per GENERATOR from the solver's per-bus values, as pypower's pfsoln reports it (case.generator_dispatch):
        # reactive power shared among the generators of a bus by range, the first ext_grid of a REF bus balancing it
        share = generator_dispatch(net, case)

        def bus_q_mvar(i):
            return r['s_ref'][ordinal[i], 1] * base if case.bus_type[i] == REF else r['q_gen'][i] * base
        eg = net['ext_grid']
        pe, qe = np.full(len(eg), np.nan), np.full(len(eg), np.nan)
        for pos, i in enumerate(share['ext_grid']['bus']):
            if i >= 0:
                pe[pos] = share['ext_grid']['p_b'][pos] * r['s_ref'][ordinal[int(i)], 0] * base
                qe[pos] = share['ext_grid']['q_a'][pos] + share['ext_grid']['q_b'][pos] * bus_q_mvar(int(i))
        net['res_ext_grid'] = pd.DataFrame({'p_mw': pe, 'q_mvar': qe}, index=eg.index)
        # units outside the power flow (out of service, or on a de-energised bus): zero rows, as pandapower's
        # `_is_elements` mask produces them (results_bus.py write_pq_results_to_element, results_gen.py)
        bus_pos = {int(b): k for k, b in enumerate(net['bus'].index)}

        def takes_part(df):
            live = np.array([int(b) in case.bus_lookup and not np.isnan(vm[bus_pos[int(b)]]) for b in df['bus']], dtype=bool)
            if 'in_service' in df.columns:
                live &= df['in_service'].to_numpy(bool)
            return live
        for tbl in ('load', 'sgen', 'storage'):
            df = net[tbl]
            sc = df['scaling'].to_numpy(float) if 'scaling' in df.columns and len(df) else 1.0
            part = takes_part(df).astype(float) if len(df) else 1.0
            net['res_' + tbl] = pd.DataFrame({'p_mw': df['p_mw'].to_numpy(float) * sc * part,
                                              'q_mvar': df['q_mvar'].to_numpy(float) * sc * part}, index=df.index)
        # element types beyond the SimBench grids: wards (constant power + constant impedance at the solved voltage), motors,
        # series impedances (the flows at both ends from the solved voltages and the branch's own stamps)
        static = static_consumption(net)
        if 'ward' in static:
            df = net['ward']
            part = takes_part(df)
            vw = np.array([vm[bus_pos[int(b)]] for b in df['bus']])          # (the bus's voltage whether the ward is in service or not)
            v2 = np.where(part, vw, 0.0) ** 2
            net['res_ward'] = pd.DataFrame({'p_mw': part * (static['ward'][0] + df['pz_mw'].to_numpy(float) * v2),
                                            'q_mvar': part * (static['ward'][1] + df['qz_mvar'].to_numpy(float) * v2),
                                            'vm_pu': vw}, index=df.index)
        if 'xward' in static:
            # the ward's parts plus what flows into the impedance towards the internal source; the internal bus's own voltage
            df = net['xward']
            part = takes_part(df)
            vw = np.array([vm[bus_pos[int(b)]] for b in df['bus']])
            v2 = np.where(part, vw, 0.0) ** 2
            px = part * (static['xward'][0] + df['pz_mw'].to_numpy(float) * v2)
            qx = part * (static['xward'][1] + df['qz_mvar'].to_numpy(float) * v2)
            vi, ai = np.full(len(df), np.nan), np.full(len(df), np.nan)
            v = r['vm'] * np.exp(1j * r['va'])
            for k in np.flatnonzero(case.br_kind == KIND_XWARD):
                pos, f, t = int(case.br_elem[k]), int(case.f[k]), int(case.t[k])
                s_f = v[f] * np.conj(case.yff[k] * v[f] + case.yft[k] * v[t]) * base
                px[pos] += s_f.real
                qx[pos] += s_f.imag
                vi[pos], ai[pos] = r['vm'][t], np.degrees(r['va'][t])
            net['res_xward'] = pd.DataFrame({'p_mw': px, 'q_mvar': qx, 'vm_pu': vw, 'va_internal_degree': ai, 'vm_internal_pu': vi},
                                            index=df.index)
        if 'motor' in static:
            df = net['motor']
            part = takes_part(df).astype(float)
            net['res_motor'] = pd.DataFrame({'p_mw': static['motor'][0] * part, 'q_mvar': static['motor'][1] * part}, index=df.index)
        imp = _table(net, 'impedance')
        if imp is not None:
            cols = {c: np.zeros(len(imp)) for c in ('p_from_mw', 'q_from_mvar', 'p_to_mw', 'q_to_mvar', 'pl_mw', 'ql_mvar', 'i_from_ka', 'i_to_ka')}
            v = r['vm'] * np.exp(1j * r['va'])
            for k in np.flatnonzero(case.br_kind == KIND_IMPEDANCE):
                pos, f, t = int(case.br_elem[k]), int(case.f[k]), int(case.t[k])
                i_f, i_t = case.yff[k] * v[f] + case.yft[k] * v[t], case.ytf[k] * v[f] + case.ytt[k] * v[t]
                s_f, s_t = v[f] * np.conj(i_f) * base, v[t] * np.conj(i_t) * base
                cols['p_from_mw'][pos], cols['q_from_mvar'][pos], cols['p_to_mw'][pos], cols['q_to_mvar'][pos] = s_f.real, s_f.imag, s_t.real, s_t.imag
                cols['pl_mw'][pos], cols['ql_mvar'][pos] = (s_f + s_t).real, (s_f + s_t).imag
                cols['i_from_ka'][pos] = abs(i_f) * base / (np.sqrt(3.0) * case.vn_kv[f])
                cols['i_to_ka'][pos] = abs(i_t) * base / (np.sqrt(3.0) * case.vn_kv[t])
            net['res_impedance'] = pd.DataFrame(cols, index=imp.index)
        gen = expand_dclines(net)['gen']          # (the net's own generators, then two per DC line: to bus, from bus)
        sc = gen['scaling'].to_numpy(float) if 'scaling' in gen.columns and len(gen) else 1.0
        pg, qg, vg = np.zeros(len(gen)), np.zeros(len(gen)), np.zeros(len(gen))
        if len(gen):
            live = takes_part(gen)
            pset = gen['p_mw'].to_numpy(float) * sc
            for pos, b in enumerate(gen['bus'].to_numpy()):
                if live[pos]:
                    i = case.bus_lookup[int(b)]
                    pg[pos], vg[pos] = pset[pos], r['vm'][i]
                    qg[pos] = share['gen']['q_a'][pos] + share['gen']['q_b'][pos] * bus_q_mvar(i)
        n_own = len(net['gen'])
        net['res_gen'] = pd.DataFrame({'p_mw': pg[:n_own], 'q_mvar': qg[:n_own], 'vm_pu': vg[:n_own]}, index=net['gen'].index)
        dc = _table(net, 'dcline')
        if dc is not None:
            # pandapower `_get_dcline_results`: what the line takes at its from end and delivers at its to end, seen as consumption
            p_to, p_from, q_to, q_from = pg[n_own::2], pg[n_own + 1::2], qg[n_own::2], qg[n_own + 1::2]
            ends = {side: np.array([bus_pos[int(b)] for b in dc[side + '_bus']]) for side in ('from', 'to')}
            net['res_dcline'] = pd.DataFrame({
                'p_from_mw': -p_from, 'q_from_mvar': -q_from, 'p_to_mw': -p_to, 'q_to_mvar': -q_to, 'pl_mw': -(p_to + p_from),
                'vm_from_pu': vm[ends['from']], 'va_from_degree': va[ends['from']],
                'vm_to_pu': vm[ends['to']], 'va_to_degree': va[ends['to']]}, index=dc.index)


_default = None


def power_flow_solver(net, **kwargs):
    """Module-level callable with the reference's plug-in signature
    `Callable[[pandapowerNet], None]` (opf_env.py:53)."""
    global _default
    if _default is None:
        _default = BatchedPowerFlowSolver()
    return _default(net, **kwargs)
