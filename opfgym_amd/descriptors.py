"""Compilers of the two flat descriptors the library takes (include/opfx.h): `opfx_env_desc` — what `OpfEnv.__init__` wires,
opf_env.py:27-175: actuators, injections, cost rows, constraints, reward, observation — and `opfx_reset_desc` — the reset
programme (`_sampling` / `_set_simbench_state`, opf_env.py:222-372, and the environments' `_sampling` tails).

Split out of batched_env.py in round 6, no behaviour change: `DescriptorCompiler` is a mix-in of
`BatchedOpfEnv`, which owns the state these methods read and write (net, case, store, keys, constraints, reward)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi
from . import reward as reward_mod
from .case import KIND_LINE, KIND_TRAFO, KIND_TRAFO3W, REF, net_to_case, static_consumption
from .grids import factored_profile
from .store import OpsBuilder, _normal_and_clip, _truncated_normal

_POLY_COEF = {'cp0_eur': 0, 'cp1_eur_per_mw': 1, 'cp2_eur_per_mw2': 2,
              'cq0_eur': 3, 'cq1_eur_per_mvar': 4, 'cq2_eur_per_mvar2': 5}

_DISCRETE_KIND = {'closed': capi.ACT_BOOLEAN, 'in_service': capi.ACT_BOOLEAN,
                  'tap_pos': capi.ACT_INTEGER, 'step': capi.ACT_INTEGER}          # opf_env.py:476-481


def _case_all_branches_in(net, act_keys, bus_bus_open=False):
    """The plan is compiled with every branch an actuator can switch present (closed / in
    service); their per-instance state then only changes Ybus VALUES (opfx_env_desc.bmod_*).
    A bus-bus switch changes the bus SET instead: its state belongs to the plan (`_topology_variant`).  `bus_bus_open`: compile
    with every bus-bus switch ACTUATOR open — the environment that owns the batch does, so that its result bank has a row
    for every bus any topology can tell apart; a topology variant keeps the states of its net."""
    saved = []
    for unit, col, idxs in act_keys:
        if col in ('closed', 'in_service') and unit in ('switch', 'line', 'trafo') and len(idxs):
            idxs = list(idxs)
            if unit == 'switch':
                bb = [i for i in idxs if net['switch'].at[i, 'et'] == 'b']
                idxs = [i for i in idxs if net['switch'].at[i, 'et'] in ('l', 't')]
                if bb and bus_bus_open:
                    saved.append((unit, col, bb, net[unit].loc[bb, col].copy()))
                    net[unit].loc[bb, col] = False
            saved.append((unit, col, idxs, net[unit].loc[idxs, col].copy()))
            net[unit].loc[idxs, col] = True
    try:
        return net_to_case(net)
    finally:
        for unit, col, idxs, old in saved:
            net[unit].loc[idxs, col] = old


def _branch_stamps(case, k):
    y = (case.yff[k], case.yft[k], case.ytf[k], case.ytt[k])
    return [v for z in y for v in (float(np.real(z)), float(np.imag(z)))]


def _keep(lst, arr, kind):
    a = np.ascontiguousarray(arr, dtype=np.float64 if kind == 'd' else np.int32)
    lst.append(a)
    return a.ctypes.data_as(capi._pd if kind == 'd' else capi._pi)


class DescriptorCompiler:
    """See the module docstring."""

    def _step_columns(self):
        """(table, column) pairs that a descriptor of the step may name: bus injections, actuators with their range and clamp
        columns, table observations, prices, per-instance voltage set-points.  Every other per-instance column is only
        written by the reset (and readable through `table_column`)."""
        hot = {(t, c) for t in ('load', 'sgen', 'storage') for c in ('p_mw', 'q_mvar')} | {('gen', 'p_mw'), ('ext_grid', 'vm_pu'), ('gen', 'vm_pu')}
        hot |= {(t, c) for t in ('ward', 'xward', 'motor') for c in ('_p_mw', '_q_mvar')}
        for unit, col, _ in self.act_keys:
            hot |= {(unit, col)} | {(unit, pre + col) for pre in ('min_', 'max_', 'min_min_', 'max_max_')}
        hot |= {(unit, col) for unit, col, _ in self.obs_keys if not unit.startswith('res_')}
        hot |= {(t, c) for (t, c) in self.store.ranges if t in ('poly_cost', 'pwl_cost')}
        return hot

    def _build_sampling(self):
        self.ops = OpsBuilder(self.store)
        self.tables = []
        modes = {self.train_data, self.test_data}
        if modes - {'simbench', 'noisy_simbench', 'full_uniform', 'normal_around_mean', 'mixed'}:
            raise NotImplementedError(f'data distributions {modes} are not supported')
        # 'mixed' (opf_env.py:242-251): every reset draws one of the three sources per instance; the
        # ops of all three are compiled with the set of sources they run under
        self.mixed = 'mixed' in modes and 'noise_factor' not in self.sampling_params     # (:231 comes first)
        # Data source per distribution: 0 profile row (+noise), 1 uniform in the data range, 2 normal around
        # the mean (opf_env.py:231-241; a `noise_factor` in sampling_params sends EVERY distribution down the
        # profile path, :231).  Train and test distribution may differ (the reference's default is
        # test_data='simbench' whatever train_data is): the ops of each source are then compiled with the
        # source they run under, as for 'mixed', and a reset runs all its instances in the source of the
        # distribution it samples from.
        src = {'simbench': 0, 'noisy_simbench': 0, 'full_uniform': 1, 'normal_around_mean': 2}
        force0 = 'noise_factor' in self.sampling_params
        self.source_of = {d: (0 if force0 else src.get(d, 0)) for d in modes if d != 'mixed' or force0}
        self.per_source = self.mixed or len(set(self.source_of.values())) > 1
        sources = {0, 1, 2} if self.mixed else set(self.source_of.values())
        self.data_probabilities = tuple(self.sampling_params.get('data_probabilities', (0.5, 0.75, 1.0)))
        self.noise_factor = float(self.sampling_params.get('noise_factor', 0.1 if self.mixed else 0.0))   # :318 default
        if 'noisy_simbench' in modes and 'noise_factor' not in self.sampling_params:
            self.noise_factor = 0.1                                        # opf_env.py:318 default
        self.noise_distribution = self.sampling_params.get('noise_distribution', 'uniform')
        assert self.noise_distribution in ('uniform', 'normal')
        self.interpolate_steps = bool(self.sampling_params.get('interpolate_steps', False))
        self.uses_profiles = 0 in sources
        if self.uses_profiles:
            for key in self.profiles.keys():                               # opf_env.py:339-372
                df = self.profiles[key]
                if not df.shape[1]:
                    continue
                unit, col = key
                rel, typ, peak = factored_profile(self.profiles, key)
                slots = self.store.slots(unit, col, df.columns, dynamic=True)
                self.tables.append(dict(rel=rel, typ=typ, peak=peak, slot=slots,
                                        col_min=df.min().to_numpy(float), col_max=df.max().to_numpy(float)))
        if 2 in sources:                                                   # opf_env.py:286-315
            self.ops.mode_mask = 4 if self.per_source else 7
            truncated = bool(self.sampling_params.get('truncated'))
            rel = self.sampling_params.get('relative_std')
            for unit, col, idxs in self.state_keys:
                if 'res_' in unit or 'poly_cost' in unit:
                    continue
                df = self.net[unit]
                rows = self.store.rows(unit, idxs)
                sc = df['scaling'].to_numpy(float)[rows]
                hi = df[f'max_max_{col}'].to_numpy(float)[rows] / sc
                lo = df[f'min_min_{col}'].to_numpy(float)[rows] / sc
                diff = hi - lo
                std = rel * diff if rel else df[f'std_dev_{col}'].to_numpy(float)[rows]
                if truncated:                                                # :304-307
                    _truncated_normal(self.ops, unit, col, idxs, df[f'mean_{col}'].to_numpy(float)[rows],
                                      std * diff, lo, hi)
                else:
                    _normal_and_clip(self.ops, unit, col, idxs, df[f'mean_{col}'].to_numpy(float)[rows],
                                     std * diff, lo, hi)                     # (std * diff as at :312)
        if 1 in sources:
            self.ops.mode_mask = 2 if self.per_source else 7
            for unit, col, idxs in self.state_keys:                        # opf_env.py:253-284
                if 'res_' in unit:
                    continue
                df = self.net[unit]
                rows = self.store.rows(unit, idxs)
                lo = df[f'min_min_{col}' if f'min_min_{col}' in df else f'min_{col}'].to_numpy(float)[rows]
                hi = df[f'max_max_{col}' if f'max_max_{col}' in df else f'max_{col}'].to_numpy(float)[rows]
                sc = df['scaling'].to_numpy(float)[rows] if 'scaling' in df else 1.0
                self.ops.uniform(unit, col, idxs, lo, hi, sc)
        self.ops.mode_mask = 7
        self._sampling_ops(self.ops)

    # ------------------------------------------------------------------ compile
    def _branch_state_column(self, unit, col, idxs, rows, bmod, a0=0):
        """Actuator columns that change Ybus values per instance: transformer tap
        positions (one stamp table row per integer position, computed by the case builder itself),
        line/trafo switches or in_service flags (stamps or nothing) and shunts in steps (the bus's
        shunt admittance per integer step)."""
        net, c, st = self.net, self.case, self.store
        br_of = {(int(kd), int(e)): k for k, (kd, e) in enumerate(zip(c.br_kind, c.br_elem))}

        def branch(kind, pos, what):
            if (kind, int(pos)) not in br_of:
                raise ValueError(f'{what}: the element is not part of the energised grid')
            return br_of[(kind, int(pos))]
        slot0 = st.slot(unit, col, dynamic=True)
        if col == 'tap_pos' and unit == 'trafo':
            df = net['trafo']
            lo_col = 'min_min_tap_pos' if 'min_min_tap_pos' in df.columns else 'min_tap_pos'
            hi_col = 'max_max_tap_pos' if 'max_max_tap_pos' in df.columns else 'max_tap_pos'
            lo = int(np.floor(df[lo_col].loc[list(idxs)].min()))
            hi = int(np.ceil(df[hi_col].loc[list(idxs)].max()))
            saved = df['tap_pos'].copy()
            tables = {int(r): [] for r in rows}
            try:
                for pos in range(lo, hi + 1):
                    df.loc[list(idxs), 'tap_pos'] = pos
                    cp = _case_all_branches_in(net, self.act_keys, not self._topology_fixed)
                    assert cp.nbr == c.nbr
                    for r in rows:
                        tables[int(r)].append(_branch_stamps(cp, branch(KIND_TRAFO, r, 'trafo.tap_pos')))
            finally:
                net['trafo']['tap_pos'] = saved
            for r in rows:
                bmod.append(dict(branch=branch(KIND_TRAFO, r, 'trafo.tap_pos'), slot=slot0 + int(r), lo=lo,
                                 table=tables[int(r)]))
        elif col in ('closed', 'in_service') and unit in ('switch', 'line', 'trafo'):
            for j, r in enumerate(rows):
                if unit == 'switch':
                    et, elem = net['switch']['et'].iloc[int(r)], int(net['switch']['element'].iloc[int(r)])
                    if et == 'b':
                        # a bus-bus switch FUSES two buses when closed: another bus set, another plan.  Its column is a plain
                        # integer column of the store here; the instances of a step are routed, by the states of these
                        # switches, to twins of this environment compiled on that topology (`_launch_step_by_topology`)
                        self._bb_switches.append(dict(row=int(r), act=a0 + j, slot=slot0 + int(r),
                                                      index=net['switch'].index[int(r)]))
                        continue
                    if et not in ('l', 't'):
                        raise NotImplementedError(f"switch.closed: element type '{et}' is not supported")
                    tbl = 'line' if et == 'l' else 'trafo'
                    kind, pos = (KIND_LINE if et == 'l' else KIND_TRAFO), st.rows(tbl, [elem])[0]
                    k = branch(kind, pos, 'switch.closed')
                    # open: the element stays connected at its other end (a shunt there, case.py
                    # open_ended_stamps) unless a second switch is open too — asked from the case builder itself
                    sw_idx = net['switch'].index[int(r)]
                    saved = bool(net['switch'].at[sw_idx, 'closed'])
                    others = [(u, cl, [i for i in ix if not (u == 'switch' and i == sw_idx)])
                              for u, cl, ix in self.act_keys]
                    try:
                        net['switch'].at[sw_idx, 'closed'] = False
                        cp = _case_all_branches_in(net, others, not self._topology_fixed)
                    finally:
                        net['switch'].at[sw_idx, 'closed'] = saved
                    hit = [j for j, (kd, e) in enumerate(zip(cp.br_kind, cp.br_elem)) if (int(kd), int(e)) == (kind, int(pos))]
                    opened = _branch_stamps(cp, hit[0]) if hit else [0.0] * 8
                else:
                    k = branch(KIND_LINE if unit == 'line' else KIND_TRAFO, r, f'{unit}.in_service')
                    opened = [0.0] * 8
                bmod.append(dict(branch=k, slot=slot0 + int(r), lo=0, table=[opened, _branch_stamps(c, k)]))
        elif col == 'step' and unit == 'shunt':
            # a shunt in steps (opf_env.py:476-481 rounds the set-point): the bus's shunt admittance for every integer step,
            # computed by the case builder itself, as the DIFFERENCE to the compiled case (bmod_branch = -1 - bus)
            df = net['shunt']
            lo_col = 'min_min_step' if 'min_min_step' in df.columns else 'min_step'
            hi_col = 'max_max_step' if 'max_max_step' in df.columns else ('max_step' if 'max_step' in df.columns else None)
            lo = int(np.floor(df[lo_col].loc[list(idxs)].min())) if lo_col in df.columns else 0
            if hi_col is None:
                raise ValueError("('shunt', 'step') actuator: the shunt table needs max_step (or max_max_step)")
            hi = int(np.ceil(df[hi_col].loc[list(idxs)].max()))
            buses = {}
            for r in rows:
                b = int(df['bus'].iloc[int(r)])
                if b not in c.bus_lookup:
                    raise ValueError('shunt.step: the shunt is not part of the energised grid')
                if not bool(df['in_service'].iloc[int(r)] if 'in_service' in df.columns else True):
                    raise ValueError('shunt.step: the shunt is out of service')
                if c.bus_lookup[b] in buses.values():
                    raise NotImplementedError('shunt.step: two controllable shunts at one bus')
                buses[int(r)] = c.bus_lookup[b]
            saved = df['step'].copy()
            tables = {int(r): [] for r in rows}
            try:
                for pos in range(lo, hi + 1):
                    for r in rows:
                        net['shunt']['step'] = saved                         # (one shunt at a time: buses may be fused)
                        net['shunt'].loc[df.index[int(r)], 'step'] = pos
                        cp = _case_all_branches_in(net, self.act_keys, not self._topology_fixed)
                        i = buses[int(r)]
                        tables[int(r)].append([0.0] * 6 + [float(cp.gs[i] - c.gs[i]), float(cp.bs[i] - c.bs[i])])
            finally:
                net['shunt']['step'] = saved
            for r in rows:
                bmod.append(dict(branch=-1 - buses[int(r)], slot=slot0 + int(r), lo=lo, table=tables[int(r)]))
        else:
            raise NotImplementedError(f'actuator {unit}.{col} is not supported')

    def _range_source(self, unit, name, rows):
        """(slots, consts) for a range/clamp column: per-instance slot if the
        sampling programme writes it, the net's static value otherwise."""
        if (unit, name) in self.store.dynamic:
            return self.store.slot(unit, name) + rows, np.zeros(len(rows))
        return np.full(len(rows), -1), self.net[unit][name].to_numpy(float)[rows]

    def _result_index(self, unit, col, idxs):
        c = self.case
        nb, nbr = c.nb, c.nbr
        ref_buses = np.flatnonzero(c.bus_type == REF)
        nref = len(ref_buses)
        zero = 2 * nb + nbr + 2 * nref + int(ref_buses[0])      # q_gen of a REF bus is always 0
        out = []
        if unit == 'bus':
            off = {'vm_pu': 0, 'va_degree': nb}[col]
            for b in idxs:
                out.append(off + c.bus_lookup[int(b)] if int(b) in c.bus_lookup else -1)
        elif unit in ('line', 'trafo'):
            assert col == 'loading_percent'
            kind = KIND_LINE if unit == 'line' else KIND_TRAFO
            pos_to_br = {int(e): k for k, (kd, e) in enumerate(zip(c.br_kind, c.br_elem)) if kd == kind}
            for pos in self.store.rows(unit, idxs):
                out.append(2 * nb + pos_to_br[int(pos)] if int(pos) in pos_to_br else zero)
        elif unit == 'trafo3w':
            # pandapower's res_trafo3w.loading_percent = the worst of the three windings: a derived row
            # (OPFX_XRES_MAX3) over the loadings of the three branches of its star equivalent
            assert col == 'loading_percent'
            base3 = 3 * nb + nbr + 2 * nref
            for pos in self.store.rows(unit, idxs):
                br = [k for k, (kd, e) in enumerate(zip(c.br_kind, c.br_elem)) if kd == KIND_TRAFO3W and int(e) == int(pos)]
                if len(br) != 3:
                    out.append(zero)                               # out of service: 0 %
                    continue
                key = ('trafo3w', col, int(pos))
                if key not in self._xres:
                    self._new_derived_row(key)
                    self._xres[key] = (len(self._xres), capi.XRES_MAX3, 2 * nb + br[0], 2 * nb + br[1], 1.0, 2 * nb + br[2], 0.0)
                out.append(base3 + self._xres[key][0])
        elif unit == 'ext_grid':
            off = 2 * nb + nbr + (0 if col == 'p_mw' else nref)
            ordinal = {int(b): k for k, b in enumerate(ref_buses)}
            share = self._generator_shares()['ext_grid']
            for pos in self.store.rows(unit, idxs):
                bus = int(self.net.ext_grid['bus'].iloc[pos])
                if bus not in c.bus_lookup:
                    out.append(-1)
                    continue
                src = off + ordinal[c.bus_lookup[bus]]
                a, b = (0.0, float(share['p_b'][pos])) if col == 'p_mw' else (float(share['q_a'][pos]), float(share['q_b'][pos]))
                # an ext_grid alone on its bus reads the bus value; one that shares it with other generators its own share
                # (a derived row; allocated for every ext_grid of a net whose topology may fuse generator buses, so that
                # the twins of a bus-bus-switch environment number their derived rows alike)
                if (a, b) == (0.0, 1.0) and not self._shares_may_change():
                    out.append(src)
                else:
                    out.append(self._affine_row(('ext_grid', col, int(pos)), src, a, b, -1))
        elif unit == 'gen' and col == 'q_mvar':
            # res_gen.q_mvar: the generator's share of the reactive power generated at its bus (pypower pfsoln,
            # case.generator_dispatch) — a derived row, affine in the bus total, where the bus is shared; zero for a generator out of service
            share = self._generator_shares()['gen']
            ordinal = {int(b): k for k, b in enumerate(ref_buses)}
            for pos in self.store.rows(unit, idxs):
                i = int(share['bus'][pos])
                if i < 0:
                    out.append(zero)
                    continue
                src = 2 * nb + nbr + nref + ordinal[i] if c.bus_type[i] == REF else 2 * nb + nbr + 2 * nref + i
                a, b = float(share['q_a'][pos]), float(share['q_b'][pos])
                # (alone on its bus: the bus's own entry — unless another switch state may give it company, see ext_grid)
                if (a, b) == (0.0, 1.0) and not self._shares_may_change():
                    out.append(src)
                else:
                    out.append(self._affine_row(('gen', col, int(pos)), src, a, b, i))
        elif unit in ('sgen', 'load', 'storage', 'gen') and col in ('p_mw', 'q_mvar', 's_mva'):
            # res_<unit> echoes of the set-points (= table value x scaling) and their apparent power:
            # derived rows behind the solver's result bank (opfx_env_desc.xres_*), allocated on demand
            if unit == 'gen' and col != 'p_mw':
                raise NotImplementedError(f'res_gen.{col} is not in the device result bank')
            df = self.net[unit]
            base = 3 * nb + nbr + 2 * nref
            p0 = self.store.slot(unit, 'p_mw')
            q0 = self.store.slot(unit, 'q_mvar') if unit != 'gen' else None
            for pos in self.store.rows(unit, idxs):
                key = (unit, col, int(pos))
                if key not in self._xres:
                    self._new_derived_row(key)
                    sc = float(df['scaling'].iloc[pos]) if 'scaling' in df.columns else 1.0
                    kind = capi.XRES_S if col == 's_mva' else capi.XRES_P
                    psl = (q0 if col == 'q_mvar' else p0) + int(pos)
                    qsl = q0 + int(pos) if col == 's_mva' else -1
                    self._xres[key] = (len(self._xres), kind, psl, qsl, sc, 0, 0.0)
                out.append(base + self._xres[key][0])
        else:
            raise NotImplementedError(f'result column res_{unit}.{col} is not in the device result bank')
        return np.array(out, dtype=np.int64)

    def _generator_shares(self):
        """`case.generator_dispatch` of this environment's net and case (cached per compiled case)."""
        cached = getattr(self, '_gen_shares', None)
        if cached is None or cached[0] is not self.case:
            from .case import generator_dispatch
            cached = self._gen_shares = (self.case, generator_dispatch(self.net, self.case))
        return cached[1]

    def _shares_may_change(self):
        """A net with bus-bus switches and more than one generator row: another switch state may put generators on one bus."""
        sw = self.net['switch'] if 'switch' in self.net else None
        has_bb = sw is not None and len(sw) and any(str(v) == 'b' for v in sw['et'])
        return bool(has_bb) and len(self.net['gen']) + len(self.net['ext_grid']) > 1

    def _new_derived_row(self, key):
        """Derived rows exist in the result bank only if the compiled environment asked for them (an observation, a
        constraint, an objective term or a cost row reads them): a request after compilation has no column to point at."""
        if getattr(self, '_xres_frozen', False):
            raise KeyError(f'res_{key[0]}.{key[1]} (row {key[2]}) is not in this environment\'s result bank: derived rows are '
                           f'compiled in when an observation, constraint, objective term or cost row reads them')

    def _affine_row(self, key, src, a, b, bus):
        """Result index of the derived row `a + b * result[src]` (OPFX_XRES_AFFINE; 0 while `bus` is de-energised)."""
        if key not in self._xres:
            self._new_derived_row(key)
            self._xres[key] = (len(self._xres), capi.XRES_AFFINE, int(src), int(bus), float(b), 0, float(a))
        nref = int((self.case.bus_type == REF).sum())
        return 3 * self.case.nb + self.case.nbr + 2 * nref + self._xres[key][0]

    def _create_env(self):
        net, c, st = self.net, self.case, self.store
        self._xres, self._xres_frozen = {}, False
        nb, base = c.nb, c.base_mva
        keep = []
        d = capi.EnvDesc()
        # ---- observation sources first (may register static columns) ------------
        okind, oidx, self.obs_segments = [], [], []
        for unit, col, idxs in self.obs_keys:
            if unit.startswith('res_'):
                ridx = self._result_index(unit[4:], col, idxs)
                if (ridx < 0).any():
                    raise ValueError(f'observation {unit}.{col} touches a de-energised element')
                okind += [capi.SRC_RESULT] * len(ridx)
                oidx += ridx.tolist()
                self.obs_segments.append(len(ridx))
            else:
                sl = st.slots(unit, col, idxs)
                okind += [capi.SRC_X] * len(sl)
                oidx += sl.tolist()
                self.obs_segments.append(len(sl))
        # ---- actions (opf_env.py:421-491) ------------------------------------------
        a_slot, a_sc, lo_s, hi_s, lo_c, hi_c = [], [], [], [], [], []
        a_kind, bmod = [], []
        a_part = []            # 1.0: the unit takes part in the power flow (its res_ row echoes the set-point), 0.0: it does not
        self._bb_switches = []
        cl_s, ch_s, cl_c, ch_c = [], [], [], []
        clamp = (not self.autoscale_actions) or bool(self.diff_action_step_size)
        for unit, col, idxs in self.act_keys:
            if len(idxs) == 0:
                continue
            df = net[unit]
            rows = st.rows(unit, idxs)
            a_kind += [_DISCRETE_KIND.get(col, capi.ACT_CONTINUOUS)] * len(rows)
            if col in _DISCRETE_KIND:
                self._branch_state_column(unit, col, idxs, rows, bmod, a0=len(a_slot))
            a_slot += (st.slot(unit, col) + rows).tolist()
            a_sc += (df['scaling'].to_numpy(float)[rows] if 'scaling' in df.columns
                     else np.ones(len(rows))).tolist()
            live = np.ones(len(rows))
            if unit in ('load', 'sgen', 'storage', 'gen') and col in ('p_mw', 'q_mvar'):
                # (pandapower reports zero power for a unit out of service or on a bus outside the power flow,
                #  results_bus.py / results_gen.py: what `get_current_actions(from_results_table=True)` reads, opf_env.py:574)
                on = df['in_service'].to_numpy(bool)[rows] if 'in_service' in df.columns else np.ones(len(rows), bool)
                live = np.array([float(o and int(b) in c.bus_lookup) for o, b in zip(on, df['bus'].to_numpy()[rows])])
            a_part += live.tolist()
            pre_lo, pre_hi = ('min_', 'max_') if self.autoscale_actions else ('min_min_', 'max_max_')
            s, v = self._range_source(unit, pre_lo + col, rows); lo_s += s.tolist(); lo_c += v.tolist()
            s, v = self._range_source(unit, pre_hi + col, rows); hi_s += s.tolist(); hi_c += v.tolist()
            for name, ss, cc in ((f'min_{col}', cl_s, cl_c), (f'max_{col}', ch_s, ch_c)):
                if clamp and (name in df.columns or (unit, name) in st.dynamic):
                    s, v = self._range_source(unit, name, rows)
                    ss += s.tolist(); cc += v.tolist()
                else:
                    ss += [-2] * len(rows); cc += [0.0] * len(rows)
        na = len(a_slot)
        if self._bb_switches:
            # what `_apply_actions` needs for the bus-bus switch columns alone (their state decides the topology BEFORE the
            # launch): range and clamp limits, constants of the switch table (opf_env.py:439-470)
            cols = [sw['act'] for sw in self._bb_switches]
            if any(lo_s[c] >= 0 or hi_s[c] >= 0 or cl_s[c] >= 0 or ch_s[c] >= 0 for c in cols):
                raise NotImplementedError('bus-bus switch actuators with sampled (per-instance) limits')
            nan = float('nan')
            self._bb_act = dict(cols=cols, slots=[sw['slot'] for sw in self._bb_switches],
                                lo=[lo_c[c] for c in cols], hi=[hi_c[c] for c in cols], sc=[a_sc[c] for c in cols],
                                cl=[cl_c[c] if cl_s[c] == -1 else nan for c in cols],
                                ch=[ch_c[c] if ch_s[c] == -1 else nan for c in cols])
            if len(cols) > 6:
                raise NotImplementedError(f'{len(cols)} bus-bus switch actuators: up to 6 (64 topologies) are supported')
            if not self._topology_fixed:
                if self.on_pivot_breakdown == 'resolve':
                    raise NotImplementedError("on_pivot_breakdown='resolve' together with bus-bus switch actuators")
                if self.host_mode and self.n_minus_one_keys:
                    raise NotImplementedError('host callables under N-1 keys together with bus-bus switch actuators')
        # ---- bus injections (makeSbus) -----------------------------------------------
        plist = [[] for _ in range(nb)]
        qlist = [[] for _ in range(nb)]
        for tbl, sign, has_q in (('load', -1.0, True), ('sgen', 1.0, True), ('storage', -1.0, True),
                                 ('gen', 1.0, False)):
            df = net[tbl]
            if not len(df):
                continue
            on = df['in_service'].to_numpy(bool) if 'in_service' in df.columns else np.ones(len(df), bool)
            sc = df['scaling'].to_numpy(float) if 'scaling' in df.columns else np.ones(len(df))
            p0 = st.slot(tbl, 'p_mw')
            q0 = st.slot(tbl, 'q_mvar') if has_q else None
            for pos, b in enumerate(df['bus'].to_numpy()):
                if on[pos] and int(b) in c.bus_lookup:
                    i = c.bus_lookup[int(b)]
                    plist[i].append((p0 + pos, sign * sc[pos] / base))
                    if has_q:
                        qlist[i].append((q0 + pos, sign * sc[pos] / base))

        # (wards' constant-power part and motors: consumption that nothing samples or actuates — one computed column each)
        for tbl, (pv, qv) in static_consumption(net).items():
            p0, q0 = st.computed(tbl, '_p_mw', pv), st.computed(tbl, '_q_mvar', qv)
            for pos, b in enumerate(net[tbl]['bus'].to_numpy()):
                if int(b) in c.bus_lookup:
                    plist[c.bus_lookup[int(b)]].append((p0 + pos, -1.0 / base))
                    qlist[c.bus_lookup[int(b)]].append((q0 + pos, -1.0 / base))

        def csr(lists):
            ptr = np.zeros(nb + 1, dtype=np.int32)
            ptr[1:] = np.cumsum([len(l) for l in lists])
            return ptr, [e[0] for l in lists for e in l], [e[1] for l in lists for e in l]
        pp_, ps_, pc_ = csr(plist)
        qp_, qs_, qc_ = csr(qlist)
        qg_lo = np.full(nb, -np.inf)
        qg_hi = np.full(nb, np.inf)
        gen = net['gen']
        if len(gen) and 'min_q_mvar' in gen.columns:
            acc_lo, acc_hi, has = np.zeros(nb), np.zeros(nb), np.zeros(nb, bool)
            on = gen['in_service'].to_numpy(bool)
            for pos, b in enumerate(gen['bus'].to_numpy()):
                if on[pos] and int(b) in c.bus_lookup:
                    i = c.bus_lookup[int(b)]
                    lo, hi = float(gen['min_q_mvar'].iloc[pos]), float(gen['max_q_mvar'].iloc[pos])
                    acc_lo[i] += -np.inf if np.isnan(lo) else lo
                    acc_hi[i] += np.inf if np.isnan(hi) else hi
                    has[i] = True
            qg_lo[has], qg_hi[has] = acc_lo[has] / base, acc_hi[has] / base
        # ---- costs (objective.py:6-87) -----------------------------------------------
        ref_buses = np.flatnonzero(c.bus_type == REF)
        ref_ord = {int(b): k for k, b in enumerate(ref_buses)}

        cost_pres, cost_qres = [], []         # per cost row: derived rows replacing the per-bus values it reads (-1: none)

        def own_share(unit, col, pos, default):
            """Result index of `res_<unit>.<col>` of one ext_grid / generator where it is not the per-bus value `default`
            the cost row reads anyway (a unit that shares its bus, or takes no part in the power flow), else -1."""
            ridx = int(self._result_index(unit, col, [net[unit].index[pos]])[0])
            return -1 if ridx == default else ridx

        def cost_source(et, element):
            pos = int(st.rows(et, [element])[0])
            off_ref = 2 * nb + c.nbr
            if et == 'ext_grid':
                k = ref_ord[c.bus_lookup[int(net.ext_grid['bus'].iloc[pos])]]
                cost_pres.append(own_share('ext_grid', 'p_mw', pos, off_ref + k))
                cost_qres.append(own_share('ext_grid', 'q_mvar', pos, off_ref + len(ref_buses) + k))
                return capi.COST_EXT_GRID, k, -1, 1.0, -1
            sc = float(net[et]['scaling'].iloc[pos]) if 'scaling' in net[et].columns else 1.0
            if et == 'gen':
                # (a generator out of service, or on a bus outside the compiled case, reports zero power: results_gen.py)
                bus = c.bus_lookup.get(int(net.gen['bus'].iloc[pos]), -1)
                if bus < 0 or ('in_service' in net.gen.columns and not bool(net.gen['in_service'].iloc[pos])):
                    sc, bus = 0.0, (bus if bus >= 0 else int(ref_buses[0]))
                cost_pres.append(-1)
                cost_qres.append(own_share('gen', 'q_mvar', pos, off_ref + 2 * len(ref_buses) + bus))
                return capi.COST_GEN, bus, st.slot('gen', 'p_mw') + pos, sc, -1
            cost_pres.append(-1); cost_qres.append(-1)
            # (a unit on a bus that is not part of the compiled case — permanently de-energised — or out of
            #  service reports zero power, results_bus.py: its row keeps the constant term only)
            bus = c.bus_lookup.get(int(net[et]['bus'].iloc[pos]), -1)
            if bus < 0 or ('in_service' in net[et].columns and not bool(net[et]['in_service'].iloc[pos])):
                sc = 0.0
            return capi.COST_UNIT, st.slot(et, 'p_mw') + pos, st.slot(et, 'q_mvar') + pos, sc, bus
        poly, pwl = net['poly_cost'], net['pwl_cost']
        if self.objective_terms or self.host_objective is not None:   # objective_function replaces get_pandapower_costs (opf_env.py:80-84)
            poly, pwl = poly.iloc[:0], pwl.iloc[:0]
        ck, cp, cq, cs, coef, is_q, cbus = [], [], [], [], [], [], []
        for _, row in poly.iterrows():
            k, pi, qi, sc, bus = cost_source(row['et'], row['element'])
            ck.append(k); cp.append(pi); cq.append(qi); cs.append(sc); cbus.append(bus)
            coef += [float(row[n]) for n in _POLY_COEF]
        nseg = min((len(p) for p in pwl['points']), default=0) if len(pwl) else 0      # defect D9
        for _, row in pwl.iterrows():
            k, pi, qi, sc, bus = cost_source(row['et'], row['element'])
            ck.append(k); cp.append(pi); cq.append(qi); cs.append(sc); cbus.append(bus)
            is_q.append(0 if row['power_type'] == 'p' else 1)
            for sgm in row['points'][:nseg]:
                coef += [float(v) for v in sgm]
        price_slot, price_coef = [], []
        for (tbl, col) in sorted(st.dynamic):
            if tbl == 'poly_cost' and col in _POLY_COEF:
                for r in range(len(poly)):
                    price_slot.append(st.slot(tbl, col) + r); price_coef.append(r * 6 + _POLY_COEF[col])
            elif tbl == 'pwl_cost' and (col in getattr(self, 'pwl_price_columns', {}) or
                                        (col == 'cp1_eur_per_mw' and not getattr(self, 'pwl_price_columns', None))):
                # per-instance segment prices: by default the sampled price is the price of
                # segment 0 (eco_dispatch.py:119-123); environments may name one column per segment
                seg = getattr(self, 'pwl_price_columns', {}).get(col, 0)
                for r in range(len(pwl)):
                    price_slot.append(st.slot(tbl, col) + r)
                    price_coef.append(len(poly) * 6 + (r * nseg + seg) * 3 + 2)
        # ---- constraints (constraints.py:70-128) ---------------------------------------
        con_ptr, con_src, con_min, con_max = [0], [], [], []
        c_as, c_pf, c_pp, c_cp, c_wc = [], [], [], [], []
        for con in self.device_constraints:
            lo, hi = con.boundaries(net)
            ridx = self._result_index(con.unit_type, con.values_column, net[con.unit_type].index)
            for r, l, h in zip(ridx, lo, hi):
                if r >= 0 and not (np.isnan(l) and np.isnan(h)):
                    con_src.append(int(r)); con_min.append(l); con_max.append(h)
            con_ptr.append(len(con_src))
            c_as.append(con.autoscale_factor(net)); c_pf.append(con.penalty_factor)
            c_pp.append(con.penalty_power); c_cp.append(con.violation_count_penalty)
            c_wc.append(int(bool(con.only_worst_case_violations)))
        # ---- N-1 list (security_constrained.py:44-50) -------------------------------------
        cont, cont_pos = [], []
        for unit, column, idxs in self.n_minus_one_keys:
            kind = {'line': KIND_LINE, 'trafo': KIND_TRAFO}[unit]
            pos_to_br = {int(e): k for k, (kd, e) in enumerate(zip(c.br_kind, c.br_elem)) if kd == kind}
            for pos in st.rows(unit, idxs):
                cont_pos.append(pos_to_br.get(int(pos), -1))
                if int(pos) in pos_to_br:          # already out of service -> skipped (:46-48)
                    cont.append(pos_to_br[int(pos)])
        self.contingencies = cont
        self._contingency_positions = cont_pos     # (per element of the N-1 keys: its case branch, -1 = not energised)
        # ---- fill the descriptor -------------------------------------------------------------
        self.nx = st.n
        d.nx = st.n
        d.pinj_ptr, d.pinj_slot, d.pinj_coef = _keep(keep, pp_, 'i'), _keep(keep, ps_, 'i'), _keep(keep, pc_, 'd')
        d.qinj_ptr, d.qinj_slot, d.qinj_coef = _keep(keep, qp_, 'i'), _keep(keep, qs_, 'i'), _keep(keep, qc_, 'd')
        self.n_inj = int(len(ps_) + len(qs_))
        d.qg_min, d.qg_max = _keep(keep, qg_lo, 'd'), _keep(keep, qg_hi, 'd')
        d.na = na
        d.act_slot, d.act_scaling = _keep(keep, a_slot, 'i'), _keep(keep, a_sc, 'd')
        d.act_lo_slot, d.act_hi_slot = _keep(keep, lo_s, 'i'), _keep(keep, hi_s, 'i')
        d.act_lo_const, d.act_hi_const = _keep(keep, lo_c, 'd'), _keep(keep, hi_c, 'd')
        d.clamp_lo_slot, d.clamp_hi_slot = _keep(keep, cl_s, 'i'), _keep(keep, ch_s, 'i')
        d.clamp_lo_const, d.clamp_hi_const = _keep(keep, cl_c, 'd'), _keep(keep, ch_c, 'd')
        d.clamp_enabled = int(clamp) | (int(not self.autoscale_actions) << 1)
        d.diff_action_step_size = float(self.diff_action_step_size or 0.0)
        d.clipped_action_penalty = float(self.clipped_action_penalty or 0.0)
        d.npoly, d.npwl, d.nseg = len(poly), len(pwl), nseg
        d.cost_kind, d.cost_pidx, d.cost_qidx = _keep(keep, ck, 'i'), _keep(keep, cp, 'i'), _keep(keep, cq, 'i')
        d.cost_scale, d.pwl_is_q, d.cost_coef = _keep(keep, cs, 'd'), _keep(keep, is_q, 'i'), _keep(keep, coef, 'd')
        d.cost_bus = _keep(keep, cbus, 'i')
        if any(v >= 0 for v in cost_pres + cost_qres):
            d.cost_pres, d.cost_qres = _keep(keep, cost_pres, 'i'), _keep(keep, cost_qres, 'i')
        d.nprice = len(price_slot)
        d.price_slot, d.price_coef = _keep(keep, price_slot, 'i'), _keep(keep, price_coef, 'i')
        d.nc = len(self.device_constraints)
        d.con_ptr, d.con_src = _keep(keep, con_ptr, 'i'), _keep(keep, con_src, 'i')
        d.con_min, d.con_max = _keep(keep, con_min, 'd'), _keep(keep, con_max, 'd')
        d.con_autoscale, d.con_penalty_factor = _keep(keep, c_as, 'd'), _keep(keep, c_pf, 'd')
        d.con_penalty_power, d.con_count_penalty = _keep(keep, c_pp, 'd'), _keep(keep, c_cp, 'd')
        d.con_worst_case = _keep(keep, c_wc, 'i')
        # a reward object that overrides one of the reference's extension points (adjust_objective, ...) or is not one
        # of this package's classes cannot be expressed as kernel parameters: the kernel then computes a plain
        # summation (unused) and the host finishes the reward with the user's object (host_fallback.py)
        self.host_reward = not reward_mod.runs_on_device(self.reward_function)
        if self.host_reward:
            reward_mod.check_host_reward(self.reward_function)
        rf = reward_mod.Summation() if self.host_reward else self.reward_function
        d.reward_kind = rf.KIND
        d.penalty_weight = np.nan if rf.penalty_weight is None else float(rf.penalty_weight)
        d.clip_lo, d.clip_hi = (np.nan, np.nan) if not rf.clip_range else map(float, rf.clip_range)
        sp = rf.scaling_params
        d.objective_factor, d.objective_bias = float(sp['objective_factor']), float(sp['objective_bias'])
        d.penalty_factor, d.penalty_bias = float(sp['penalty_factor']), float(sp['penalty_bias'])
        d.valid_reward, d.invalid_penalty = float(rf.valid_reward), float(rf.invalid_penalty)
        d.invalid_objective_share = float(rf.invalid_objective_share)
        d.diff_objective = int(bool(self.diff_objective))
        d.nobs = len(oidx)
        d.obs_kind, d.obs_idx = _keep(keep, okind, 'i'), _keep(keep, oidx, 'i')
        d.steps_per_episode = int(self.steps_per_episode)
        d.n_cont = len(cont)
        d.cont_branch = _keep(keep, cont, 'i')
        d.not_converged_penalty = float(self.not_converged_penalty)
        d.act_kind = _keep(keep, a_kind, 'i')
        # per-instance voltage set-points: ext_grid.vm_pu / gen.vm_pu columns that the sampling writes
        vset = np.full(nb, -1, dtype=np.int32)
        for tbl in ('ext_grid', 'gen'):
            if (tbl, 'vm_pu') in st.dynamic and len(net[tbl]):
                s0 = st.slot(tbl, 'vm_pu')
                for pos, b in enumerate(net[tbl]['bus'].to_numpy()):
                    if int(b) in c.bus_lookup and vset[c.bus_lookup[int(b)]] < 0:
                        vset[c.bus_lookup[int(b)]] = s0 + pos
        if (vset >= 0).any():
            d.vset_slot = _keep(keep, vset, 'i')
        q_idx, q_tgt, q_w = [], [], []
        for f in self.objective_terms:
            idxs = net[f.unit].index if f.idxs is None else f.idxs
            ridx = self._result_index(f.unit, f.column, idxs)
            if (ridx < 0).any():
                raise ValueError(f'objective term on res_{f.unit}.{f.column} touches a de-energised element')
            q_idx += ridx.tolist(); q_tgt += [f.target] * len(ridx); q_w += [f.weight] * len(ridx)
        d.n_qterm = len(q_idx)
        xr_ = sorted(self._xres.values())
        d.n_xres = len(xr_)
        if xr_:
            d.xres_kind, d.xres_p = _keep(keep, [v[1] for v in xr_], 'i'), _keep(keep, [v[2] for v in xr_], 'i')
            d.xres_q, d.xres_scale = _keep(keep, [v[3] for v in xr_], 'i'), _keep(keep, [v[4] for v in xr_], 'd')
            d.xres_r = _keep(keep, [v[5] for v in xr_], 'i')
            d.xres_offset = _keep(keep, [v[6] for v in xr_], 'd')
        if q_idx:
            d.qterm_idx, d.qterm_target, d.qterm_weight = _keep(keep, q_idx, 'i'), _keep(keep, q_tgt, 'd'), _keep(keep, q_w, 'd')
        d.n_bmod = len(bmod)
        if bmod:
            ptr = np.cumsum([0] + [len(b['table']) for b in bmod])
            d.bmod_branch = _keep(keep, [b['branch'] for b in bmod], 'i')
            d.bmod_slot = _keep(keep, [b['slot'] for b in bmod], 'i')
            d.bmod_lo = _keep(keep, [b['lo'] for b in bmod], 'i')
            d.bmod_n = _keep(keep, [len(b['table']) for b in bmod], 'i')
            d.bmod_ptr = _keep(keep, ptr[:-1], 'i')
            d.bmod_y = _keep(keep, np.concatenate([np.asarray(b['table'], float).ravel() for b in bmod]), 'd')
        self.branch_state_columns = bmod
        for var in getattr(self, '_topology_variants', {}).values():
            var.close()
        self._topology_variants = {}
        if getattr(self, '_env_handle_base_only', None) is not None:
            capi.lib().opfx_env_destroy(self._env_handle_base_only)
            self._env_handle_base_only = None
        if self._env_handle is not None:
            capi.lib().opfx_env_destroy(self._env_handle)
            self._env_handle = None
        h = C.c_void_p()
        capi.check(capi.lib().opfx_env_create(self.ctx.handle, C.byref(d), C.byref(h)), 'opfx_env_create')
        self._env_handle = h
        self._env_desc, self._env_desc_keep = d, keep        # (the rescue environments of on_pivot_breakdown reuse them)
        if getattr(self, '_env_handle_base_only', None) is not None:
            capi.lib().opfx_env_destroy(self._env_handle_base_only)
        self._env_handle_base_only = None
        if self.host_mode and cont:
            n_cont, d.n_cont = d.n_cont, 0                   # the same environment without its contingency list
            h0 = C.c_void_p()
            try:
                capi.check(capi.lib().opfx_env_create(self.ctx.handle, C.byref(d), C.byref(h0)), 'opfx_env_create (base case only)')
            finally:
                d.n_cont = n_cont
            self._env_handle_base_only = h0
        self._drop_rescue_envs()
        self.n_obs_raw = len(oidx)
        self.n_constraints = len(self.constraints)
        self.n_device_constraints = len(self.device_constraints)
        self._host_finisher = None
        if self.host_mode or self.host_reward:
            from .host_fallback import HostFinisher
            self._host_finisher = HostFinisher(self, self.host_objective, self._host_constraints, self._constraint_order)
        self.n_results = 3 * nb + c.nbr + 2 * len(ref_buses) + len(self._xres)
        self._xres_frozen = True
        t = self.torch
        as_i = lambda v: t.as_tensor(np.asarray(v, dtype=np.int64), device=self.device)
        as_d = lambda v: t.as_tensor(np.asarray(v, dtype=np.float64), device=self.device)
        self._act_desc = dict(slot=as_i(a_slot), scaling=as_d(a_sc), lo_slot=as_i(lo_s), hi_slot=as_i(hi_s),
                              lo_const=as_d(lo_c), hi_const=as_d(hi_c), part=as_d(a_part))
        self._set_reset()

    def _set_reset(self):
        keep = []
        st = self.store
        template = st.row_template()
        consts = [template]
        off = len(template)
        tabs = (capi.ProfileDesc * max(1, len(self.tables)))()
        self.n_noise = 0
        for k, t in enumerate(self.tables):
            tabs[k].struct_size = capi.C.sizeof(capi.ProfileDesc)
            tabs[k].n_steps, tabs[k].n_types = t['rel'].shape
            tabs[k].n_cols = len(t['typ'])
            tabs[k].rel, tabs[k].typ = _keep(keep, t['rel'], 'd'), _keep(keep, t['typ'], 'i')
            tabs[k].peak, tabs[k].slot = _keep(keep, t['peak'], 'd'), _keep(keep, t['slot'], 'i')
            tabs[k].col_min, tabs[k].col_max = _keep(keep, t['col_min'], 'd'), _keep(keep, t['col_max'], 'd')
            self.n_noise += len(t['typ'])
        code, dst, a, n, c0, c1, c2 = [], [], [], [], [], [], []
        for op in self.ops.ops:
            code.append(op[0]); dst.append(op[1]); a.append(op[2]); n.append(op[3])
            for vec, lst in ((op[4], c0), (op[5], c1), (op[6], c2)):
                if vec is None:
                    lst.append(-1)
                else:
                    lst.append(off); consts.append(vec); off += len(vec)
        consts = np.concatenate(consts) if consts else np.zeros(0)
        r = capi.ResetDesc()
        r.n_tables, r.tables = len(self.tables), tabs
        r.n_ops = len(code)
        r.op_code, r.op_dst, r.op_a, r.op_n = (_keep(keep, v, 'i') for v in (code, dst, a, n))
        r.op_c0, r.op_c1, r.op_c2 = (_keep(keep, v, 'i') for v in (c0, c1, c2))
        r.n_consts, r.consts = len(consts), _keep(keep, consts, 'd')
        r.n_uniform = self.ops.n_uniform
        r.n_normal = self.ops.n_normal
        if self.per_source:
            r.op_mode = _keep(keep, [op[7] for op in self.ops.ops], 'i')
        r.init_off = 0
        capi.check(capi.lib().opfx_env_set_reset(self._env_handle, C.byref(r)), 'opfx_env_set_reset')
        self.n_uniform = self.ops.n_uniform
        self.n_normal = self.ops.n_normal

    # ------------------------------------------------------------------ buffers
