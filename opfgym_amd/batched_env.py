"""Batched OPF environment: B independent instances of one grid per call.

Host-side mirror of `/root/reference/opfgym/opf_env.py` (`OpfEnv`) and
`security_constrained.py` (`SecurityConstrainedOpfEnv`): same constructor
arguments and gymnasium-shaped `reset()/step()` API, but every call handles a
whole batch and all numerics run on the GPU through libopfx (one kernel launch
per `step`, one per `reset`).  This file only *compiles* the problem definition
(action/observation keys, constraints, cost tables, reward, sampling programme)
into the flat descriptors of `include/opfx.h` and moves tensors; it contains no
power-flow or reward arithmetic and there is no CPU fallback.

Per-instance state is a column store x[B, nx] (torch CUDA tensor): one slot per
row of every per-instance table column the reference net would hold
(`net.load.p_mw`, `net.sgen.max_q_mvar`, `net.poly_cost.cq2_eur_per_mvar2`, …),
holding TABLE values exactly as the reference net does (i.e. before `scaling`).
"""
from __future__ import annotations

import copy
import logging
import ctypes as C

import numpy as np

from . import capi
from . import constraints as constraints_mod
from . import reward as reward_mod
from .case import KIND_LINE, KIND_TRAFO, KIND_TRAFO3W, PQ, PV, REF, net_to_case
from .grids import factored_profile
from .simbench_build import define_test_train_split, get_simbench_time_observation

_POLY_COEF = {'cp0_eur': 0, 'cp1_eur_per_mw': 1, 'cp2_eur_per_mw2': 2,
              'cq0_eur': 3, 'cq1_eur_per_mvar': 4, 'cq2_eur_per_mvar2': 5}


class ColumnStore:
    """Slot allocator for x: (table, column) -> contiguous range over the rows
    of that table, with the net's current values as the row template."""

    def __init__(self, net):
        self.net = net
        self.ranges = {}
        self.template = []
        self.n = 0
        self.dynamic = set()

    def slot(self, table, col, dynamic=False):
        key = (table, col)
        if key not in self.ranges:
            tbl = self.net[table]
            n = len(tbl)
            if col in tbl.columns:
                vals = np.array([float(v) if v is not None else np.nan
                                 for v in tbl[col].to_numpy()], dtype=float) if n else np.zeros(0)
            else:
                vals = np.zeros(n)
            self.ranges[key] = (self.n, n)
            self.template.append(vals)
            self.n += n
        if dynamic:
            self.dynamic.add(key)
        return self.ranges[key][0]

    def rows(self, table, idxs):
        pos = self.net[table].index.get_indexer(np.asarray(idxs))
        if (pos < 0).any():
            raise KeyError(f'index not in net.{table}: {np.asarray(idxs)[pos < 0]}')
        return pos

    def slots(self, table, col, idxs, dynamic=False):
        return self.slot(table, col, dynamic) + self.rows(table, idxs)

    def row_template(self):
        return np.concatenate(self.template) if self.template else np.zeros(0)


class OpsBuilder:
    """Collects the `_sampling` tail of an environment as vector ops on x
    (see OPFX_OP_* in include/opfx.h)."""

    def __init__(self, store: ColumnStore):
        self.store = store
        self.ops = []          # (code, dst, a, n, c0, c1, c2) with c* numpy vectors or None
        self.n_uniform = 0
        self.uniform_runs = []     # (first column, count, source mask) of every uniform op
        self.n_normal = 0
        self.mode_mask = 7     # data sources under which the ops added next run ('mixed' sampling)

    def _emit(self, code, dst, a, c0=None, c1=None, c2=None):
        """dst/a: arrays of slots; split into runs where both are contiguous."""
        dst = np.asarray(dst, dtype=np.int64)
        a = np.asarray(a, dtype=np.int64)
        n = len(dst)
        if n == 0:
            return
        cut = np.flatnonzero((np.diff(dst) != 1) | (np.diff(a) != 1)) + 1
        for s, e in zip(np.r_[0, cut], np.r_[cut, n]):
            sl = slice(s, e)
            self.ops.append((code, int(dst[s]), int(a[s]), int(e - s),
                             None if c0 is None else np.broadcast_to(np.asarray(c0, float), (n,))[sl].copy(),
                             None if c1 is None else np.broadcast_to(np.asarray(c1, float), (n,))[sl].copy(),
                             None if c2 is None else np.broadcast_to(np.asarray(c2, float), (n,))[sl].copy(),
                             self.mode_mask))

    def _all(self, table, col, rows=None, dynamic=False):
        base = self.store.slot(table, col, dynamic)
        n = len(self.store.net[table])
        rows = np.arange(n) if rows is None else np.asarray(rows)
        return base + rows

    def set_const(self, table, col, values, rows=None):
        dst = self._all(table, col, rows, True)
        self._emit(capi.OP_SET_CONST, dst, dst, c0=values)

    def affine(self, table, col, src_col, c0, c1, rows=None):
        dst = self._all(table, col, rows, True)
        self._emit(capi.OP_AFFINE, dst, self._all(table, src_col, rows), c0=c0, c1=c1)

    def sqrt_diff(self, table, col, src_col, c0, rows=None):
        dst = self._all(table, col, rows, True)
        self._emit(capi.OP_SQRT_DIFF, dst, self._all(table, src_col, rows), c0=c0)

    def div(self, table, col, src_col, c0, rows=None):
        dst = self._all(table, col, rows, True)
        self._emit(capi.OP_DIV, dst, self._all(table, src_col, rows), c0=c0)

    def neg(self, table, col, src_col, rows=None):
        dst = self._all(table, col, rows, True)
        self._emit(capi.OP_NEG, dst, self._all(table, src_col, rows))

    def uniform(self, table, col, idxs, lo, hi, scale=1.0):
        """opf_env.py:266-284 `_sample_from_range`: one U[lo,hi] draw per row,
        divided by `scale`; consumes len(idxs) draws of the instance's draw
        vector, in order."""
        rows = self.store.rows(table, idxs)
        dst = self._all(table, col, rows, True)
        src = self.n_uniform + np.arange(len(rows))
        self.uniform_runs.append((self.n_uniform, len(rows), self.mode_mask))
        self.n_uniform += len(rows)
        self._emit(capi.OP_UNIFORM, dst, src, c0=lo, c1=hi, c2=scale)

    def uniform_columns(self, source: int):
        """Columns of the [B, n_uniform] draw matrix that a reset under data source `source` consumes, in
        the order the reference would draw them (the matrix has one fixed column per op whatever the
        source; the reference draws sequentially and only what the source needs)."""
        cols = [np.arange(s, s + n) for s, n, mask in self.uniform_runs if (mask >> source) & 1]
        return np.concatenate(cols) if cols else np.zeros(0, dtype=np.int64)


def _truncated_normal(ops, table, col, idxs, mean, scale, a, b):
    """opf_env.py:306-309: `scipy.stats.truncnorm.rvs(min_values, max_values, mean, std * diff)` per row.
    scipy reads its first two arguments as STANDARDISED bounds, so what the reference samples is
    mean + scale * Z with Z standard normal truncated to [min_values, max_values] (the raw numbers; defect
    D14, reproduced).  scipy draws from its own generator, which cannot be replayed; here Z comes from the
    instance's uniform draw u by the inverse CDF scipy itself applies to its uniforms, `truncnorm.ppf(u, a, b)`,
    as a device op that works in log space (OPFX_OP_TRUNCNORM): bounds like [10, 200] — every unit above ~8 MW —
    lie so far in the upper tail that Phi(a) == Phi(b) == 1.0 in double precision, and the plain
    Phi^-1(Phi(a) + u (Phi(b) - Phi(a))) returns +inf there."""
    n = len(np.asarray(a, dtype=float))
    ops.uniform(table, col, idxs, np.zeros(n), np.ones(n), 1.0)      # u itself, into the column
    rows = ops.store.rows(table, idxs)
    dst = ops._all(table, col, rows, True)
    ops._emit(capi.OP_TRUNCNORM, dst, dst, c0=np.asarray(a, dtype=float), c1=np.asarray(b, dtype=float))
    ops._emit(capi.OP_AFFINE, dst, dst, c0=np.asarray(scale, dtype=float), c1=np.asarray(mean, dtype=float))


def _normal_and_clip(ops, table, col, idxs, mean, std, lo, hi):
    """opf_env.py:311-315: N(mean, std) per row clipped to [lo, hi]; consumes len(idxs)
    standard-normal draws of the instance's draw vector, in order."""
    rows = ops.store.rows(table, idxs)
    dst = ops._all(table, col, rows, True)
    src = ops.n_normal + np.arange(len(rows))
    ops.n_normal += len(rows)
    ops._emit(capi.OP_NORMAL, dst, src, c0=mean, c1=std)
    ops._emit(capi.OP_CLIP, dst, dst, c0=lo, c1=hi)


class Box:
    """Minimal stand-in for gymnasium.spaces.Box (gymnasium is optional): bounds as
    float64 arrays, `shape`, and `sample()`.  `BatchedOpfEnv` exposes the per-instance
    spaces; a batch of actions is [B, *shape]."""

    def __init__(self, low, high, shape=None, seed=None):
        low, high = np.asarray(low, dtype=float), np.asarray(high, dtype=float)
        if shape is None:
            shape = np.broadcast(low, high).shape
        self.low = np.broadcast_to(low, shape).copy()
        self.high = np.broadcast_to(high, shape).copy()
        self.shape = tuple(shape)
        self._rng = np.random.default_rng(seed)

    def sample(self, batch=None):
        shape = self.shape if batch is None else (batch,) + self.shape
        return self.low + (self.high - self.low) * self._rng.random(shape)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape[-len(self.shape):] == self.shape and bool(((x >= self.low) & (x <= self.high)).all())


def get_obs_and_state_space(net, obs_or_state_keys, add_time_obs=False, add_mean_obs=False, seed=None,
                            bus_wise_obs=False):
    """opf_env.py:720-803: observation/state bounds from the constraint columns of the net."""
    lows, highs = [], []
    if add_time_obs:                                                       # :728-732
        lows.append(-np.ones(6)); highs.append(np.ones(6))
    for unit_type, column, idxs in obs_or_state_keys:
        if 'res_' in unit_type:
            unit_type = unit_type[4:]                                      # :735-737
        elif 'max_' in column or 'min_' in column:
            column = column[4:]                                            # :738-740
        df = net[unit_type]
        if column == 'va_degree':                                          # :742-746
            lo, hi = np.full(len(idxs), -30.0), np.full(len(idxs), 30.0)
        else:
            try:
                lo = df[f'min_min_{column}' if f'min_min_{column}' in df.columns else f'min_{column}'] \
                    .loc[idxs].to_numpy(float)
                hi = df[f'max_max_{column}' if f'max_max_{column}' in df.columns else f'max_{column}'] \
                    .loc[idxs].to_numpy(float)
            except KeyError:                                               # :757-761 lines / trafos
                lo = np.zeros(len(idxs))
                hi = df[f'max_{column}'].loc[idxs].to_numpy(float) * 1.5
            if column == 'vm_pu' or unit_type == 'ext_grid':               # :764-768
                diff = hi - lo
                lo, hi = lo - diff * 0.75, hi + diff * 0.75
        if not ('min' in column or 'max' in column) and 'scaling' in df.columns:   # :770-778
            sc = df['scaling'].loc[idxs].to_numpy(float)
            lo, hi = lo / sc, hi / sc
        if bus_wise_obs and unit_type == 'load':                           # :780-784
            buses = sorted(set(df.bus))
            bus_col = df.bus.loc[idxs].to_numpy() if len(idxs) == len(df) else df.bus.to_numpy()
            lo = np.array([lo[bus_col == b].sum() for b in buses])
            hi = np.array([hi[bus_col == b].sum() for b in buses])
        if len(lo) > 0 and len(lo) == len(hi):
            lows.append(lo); highs.append(hi)
    if add_mean_obs:                                                       # :791-797
        start = 1 if add_time_obs else 0
        lows.append(np.array([np.mean(l) for l in lows[start:] if len(l) > 1]))
        highs.append(np.array([np.mean(h) for h in highs[start:] if len(h) > 1]))
    assert not any(np.isnan(l).any() for l in lows) and not any(np.isnan(h).any() for h in highs)
    return Box(np.concatenate(lows) if lows else np.zeros(0), np.concatenate(highs) if highs else np.zeros(0),
               seed=seed)


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


_noticed = set()


def _notice_deviations(env):
    """`from_reference` promises the reference's problem; say (once per process and setting) where the SOLVER SETTINGS
    are the fast defaults instead of the reference's, and how to get those."""
    dev = dict(env.reference_deviations)
    if dev.get('init') == 'flat' and env.init == 'flat' and not (env.case.meta.get('calc_angles') and env.plan.info['has_dc']):
        dev.pop('init')                  # pandapower's 'auto' is 'flat' on this grid too
    if not env.n_minus_one_keys:
        dev.pop('contingency_start', None)
    if not (env.solve_opts.enforce_q_lims and len(env.net.gen)):
        dev.pop('pin_point_q_ranges', None)
    key = tuple(sorted((k, str(v)) for k, v in dev.items()))
    if not dev or key in _noticed:
        return
    _noticed.add(key)
    why = {'init': "init='flat' (pandapower starts grids fed above 70 kV from a DC power flow: init='auto')",
           'contingency_start': "contingency_start='base_case' (the reference solves every contingency from scratch: 'flat')",
           'carry_over_state': "carry_over_state=False (the reference's single net carries unsampled columns over episodes, D12: True)",
           'pin_point_q_ranges': "pin_point_q_ranges=True (pypower starts a generator with min_q = max_q as a PV bus and pins it after "
                                 "the first solve: False)"}
    logging.getLogger('opfgym_amd').warning(
        'BatchedOpfEnv.from_reference: same problem and fixed point as the reference, but not its iteration path: %s. '
        'iterations[B], and converged[B] of rows next to voltage collapse, can differ; pass reference_faithful=True '
        '(or the single options) to reproduce it.', '; '.join(why[k] for k in dev))


class PowerFlowNotAvailable(Exception):
    """opf_env.py:22"""


class BatchedOpfEnv:
    """See module docstring.  Arguments as `OpfEnv.__init__` (opf_env.py:27-56)
    plus `batch_size`, `device` and, for the N-1 variant
    (security_constrained.py:21-35), `n_minus_one_keys` / `not_converged_penalty`."""

    def __init__(self, net, action_keys, observation_keys, state_keys=None, profiles=None,
                 evaluate_on='validation', steps_per_episode=1, bus_wise_obs=False,
                 reward_function='summation', reward_function_params=None, diff_objective=False,
                 add_res_obs=False, add_time_obs=False, add_act_obs=False, add_mean_obs=False,
                 train_data='simbench', test_data='simbench', sampling_params=None,
                 constraint_params=None, custom_constraints=None, autoscale_actions=True,
                 diff_action_step_size=None, clipped_action_penalty=0.0, initial_action='center',
                 objective_function=None, power_flow_solver=None, optimal_power_flow_solver=None,
                 seed=None, batch_size=1, device='cuda:0', n_minus_one_keys=None,
                 not_converged_penalty=1, tolerance=1e-8, max_iteration=10, enforce_q_lims=True,
                 defer_device=False, validate_actions=False, carry_over_state=None, copy_outputs=False,
                 contingency_start=None, init=None, jacobian_reuse_tol=0.0, resample_failed_resets=True,
                 on_pivot_breakdown='ignore', reference_faithful=False, pin_point_q_ranges=None, share_lds_slots='auto',
                 debug=None, **kwargs):
        from .objectives import QuadraticDeviation
        # reference_faithful: ONE switch for the four defaults that trade the reference's iteration path for speed.
        # Each of `init`, `contingency_start`, `carry_over_state`, `pin_point_q_ranges` left at None takes the fast default
        # ('flat', 'base_case', False, True) or, with reference_faithful=True, what the reference does: init='auto'
        # (pandapower's default, 'dc' on grids fed above 70 kV — SURVEY P1), contingency_start='flat' (every contingency is a
        # fresh runpp, security_constrained.py:53), carry_over_state=True (one net lives through all episodes, D12) and
        # pin_point_q_ranges=False (enforce_q_lims as pypower walks it: a generator with min_q = max_q, eco_dispatch.py:86-88,
        # starts as a PV bus and is pinned after the first converged solve instead of starting pinned — one more solve).
        # Converged results agree either way (same fixed point, same tolerance); `iterations`, and `converged` of rows
        # next to voltage collapse, follow the start.  An explicit value always wins.
        self.reference_faithful = bool(reference_faithful)
        # share_lds_slots: 'auto' (default) lets attach_device() switch to a plan with shared LDS slots where that brings a third
        # instance into the CU (_try_shared_slots); False never does
        assert share_lds_slots in ('auto', False), share_lds_slots
        self.share_lds_slots = share_lds_slots
        # debug: developer switches of the library (include/opfx_debug.h) for this environment's plan and context — None, a
        # dict of member names (`dict(team=2)`) or a capi.DebugOpts; the process environment is never consulted
        self.debug = capi.debug_opts(debug)
        faithful = dict(init='auto', contingency_start='flat', carry_over_state=True, pin_point_q_ranges=False)
        fast = dict(init='flat', contingency_start='base_case', carry_over_state=False, pin_point_q_ranges=True)
        given = dict(init=init, contingency_start=contingency_start, carry_over_state=carry_over_state,
                     pin_point_q_ranges=pin_point_q_ranges)
        resolved = {k: (v if v is not None else (faithful if reference_faithful else fast)[k]) for k, v in given.items()}
        init, contingency_start, carry_over_state, pin_point_q_ranges = (resolved[k] for k in given)
        #: the settings under which this environment does NOT walk the reference's own iteration path
        self.reference_deviations = {k: v for k, v in resolved.items() if v != faithful[k]}
        # (the arguments as given: a bus-bus switch actuator needs twins of this environment on other topologies)
        self._pre_init_attrs = {k: v for k, v in self.__dict__.items() if not k.startswith('_pre_init')}   # (what a subclass set before)
        self._topology_fixed = bool(kwargs.pop('_topology_fixed', False))
        self._ctor = {k: v for k, v in locals().items() if k not in ('self', 'net', 'kwargs', 'QuadraticDeviation', '__class__', 'faithful', 'fast', 'given', 'resolved')}
        self._ctor_kwargs = dict(kwargs)
        self._bb_switches, self._topology_variants = [], {}
        # on_pivot_breakdown: the block LU pivots statically (fixed elimination order, the 2x2 diagonal block of a bus as its
        # pivot), pandapower's SuperLU partially.  'ignore' (default): a row whose factorisation broke down is a failed row
        # like any other (`converged` 0; `min_pivot` ~ 0 and `min_pivot_bus` say why).  'resolve': such rows — not
        # converged AND min_pivot < 1e-8 — are solved once more ON THE GPU with a rescue plan that eliminates the
        # offending buses last (opfx_case.elim_last), where their diagonal blocks carry the Schur complement of the rest;
        # costs one host synchronisation per step (the count of such rows) and nothing else while there are none
        assert on_pivot_breakdown in ('ignore', 'resolve'), on_pivot_breakdown
        self.on_pivot_breakdown = on_pivot_breakdown
        self.pivot_rescues = 0                 # rows re-solved / recovered so far (diagnostics)
        self.pivot_rescues_recovered = 0
        self._rescue_envs = {}
        # resample_failed_resets (environments whose reset runs a power flow, opf_env.py:209-214): True re-samples the rows
        # whose power flow failed, which needs the convergence flags on the HOST after every reset (one stream
        # synchronisation per reset); False leaves such rows as they are (NaN observation, `converged` 0 in the buffers)
        # and keeps reset() asynchronous
        self.resample_failed_resets = bool(resample_failed_resets)
        terms = objective_function if isinstance(objective_function, (list, tuple)) else \
            ([objective_function] if objective_function is not None else [])
        if power_flow_solver is not None:
            raise NotImplementedError('a Python power-flow callable cannot replace the fused GPU solve; use the '
                                      'reference OpfEnv with opfgym_amd.power_flow_solver for that seam')
        # opf_env.py:80-84: objective_function replaces the pandapower cost tables.  Objects that describe
        # themselves run in the kernel; any other Python callable is evaluated on the host after the launch
        # (opfgym_amd/host_fallback.py: one call per instance and step — a compatibility path)
        self.host_objective = None
        if terms and not all(isinstance(f, QuadraticDeviation) for f in terms):
            if len(terms) != 1 or not callable(terms[0]):
                raise TypeError('objective_function: one callable(net) -> array, or opfgym_amd.objectives objects')
            self.host_objective, terms = terms[0], []
        self.objective_terms = list(terms)
        self.bus_wise_obs = bool(bus_wise_obs)
        self.net = net
        self.device_spec = device
        # opf_env.py:382 asserts on NaN actions; checking that on the host costs a device sync per
        # step, so it is opt-in here: by default a NaN action yields a failed (NaN) row instead
        self.validate_actions = bool(validate_actions)
        # False (default): every reset starts from the table template — independent episodes.  True: from
        # the instance's state at the end of its previous episode, as the reference's single net does; a
        # column that the previous data source set and the current one does not sample then carries over
        # (reference defect D12: e.g. gen.p_mw of the last profile row under train_data='mixed').
        self.carry_over_state = bool(carry_over_state)
        # True: reset()/step() return clones instead of the persistent output buffers (drop-in safe for rollout
        # loops that keep what they were handed; costs one device copy per returned tensor and call)
        self.copy_outputs = bool(copy_outputs)
        self._state_valid = False
        self.power_flow_available = False
        self._objective_is_diff = False
        self._last_host = None
        self.batch_size = int(batch_size)
        self.obs_keys = list(observation_keys)
        self.state_keys = list(state_keys) if state_keys else copy.copy(self.obs_keys)
        self.act_keys = list(action_keys)
        self.profiles = profiles
        if not profiles:
            assert 'simbench' not in test_data and 'simbench' not in train_data and not add_time_obs
        self.evaluate_on = evaluate_on
        self.train_data, self.test_data = train_data, test_data
        self.sampling_params = sampling_params or {}
        self.add_act_obs, self.add_time_obs, self.add_mean_obs = add_act_obs, add_time_obs, add_mean_obs
        if add_act_obs:                                                    # opf_env.py:92-95
            self.obs_keys.extend(self.act_keys)
        if add_res_obs is True:                                            # opf_env.py:99-118
            add_res_obs = ('voltage_magnitude', 'voltage_angle', 'line_loading', 'trafo_loading',
                           'ext_grid_power')
        if add_res_obs:
            bus_idxs = (set(net.load.bus) | set(net.sgen.bus) | set(net.gen.bus) | set(net.storage.bus))
            bus_idxs = np.sort(list(bus_idxs))
            if 'voltage_magnitude' in add_res_obs:
                self.obs_keys.append(('res_bus', 'vm_pu', bus_idxs))
            if 'voltage_angle' in add_res_obs:
                self.obs_keys.append(('res_bus', 'va_degree', bus_idxs))
            if 'line_loading' in add_res_obs:
                self.obs_keys.append(('res_line', 'loading_percent', net.line.index))
            if 'trafo_loading' in add_res_obs:
                self.obs_keys.append(('res_trafo', 'loading_percent', net.trafo.index))
            if 'ext_grid_power' in add_res_obs:
                self.obs_keys.append(('res_ext_grid', 'p_mw', net.ext_grid.index))
                self.obs_keys.append(('res_ext_grid', 'q_mvar', net.ext_grid.index))
        self.autoscale_actions = autoscale_actions
        self.diff_action_step_size = diff_action_step_size
        self.clipped_action_penalty = clipped_action_penalty
        self.initial_action = initial_action
        self.steps_per_episode = steps_per_episode
        self.pf_for_obs = any('res_' in k[0] for k in self.obs_keys) or bool(diff_objective)   # :144-153
        self.diff_objective = diff_objective
        self.test_steps, self.validation_steps, self.train_steps = define_test_train_split(**kwargs)  # :156
        if custom_constraints is None:                                     # :159-163
            self.constraints = constraints_mod.create_default_constraints(net, constraint_params or {})
        else:
            self.constraints = list(custom_constraints)
        if callable(n_minus_one_keys):             # decided on the prepared net (e.g. every non-islanding line)
            n_minus_one_keys = n_minus_one_keys(net)
        from . import host_fallback
        self._constraint_order, self._host_constraints, dev_constraints = [], [], []
        for con in self.constraints:
            if host_fallback.is_host_constraint(con):
                self._constraint_order.append(('host', len(self._host_constraints)))
                self._host_constraints.append(host_fallback.HostConstraint(con))
            else:
                self._constraint_order.append(('dev', len(dev_constraints)))
                dev_constraints.append(con)
        self.device_constraints = dev_constraints
        self.host_mode = self.host_objective is not None or bool(self._host_constraints)
        # (host callables together with N-1 contingencies: the kernel still walks the contingencies for its own constraints;
        #  for the callables every contingency is solved once more by a contingency-free twin of the environment with the
        #  branch out, `contingency_results`, and evaluated on the host — host_fallback.py)
        self.n_minus_one_keys = n_minus_one_keys or ()
        for _, column, _ in self.n_minus_one_keys:
            assert column in ('in_service', 'closed')                      # security_constrained.py:34-35
        self.not_converged_penalty = not_converged_penalty
        assert contingency_start in ('base_case', 'flat'), contingency_start
        # init: start of the Newton iteration — 'flat' (default), 'dc' (pandapower's init='dc': angles from a DC power
        # flow first), 'auto' (pandapower's default: 'dc' when voltage angles are calculated, i.e. grids fed above 70 kV)
        assert init in ('flat', 'dc', 'auto'), init
        self.init = init
        # jacobian_reuse_tol (default 0: full Newton, what pandapower does): once an iteration's mismatch is below it, the
        # later iterations of that solve keep its factorisation (chord steps; opfx_solve_opts.jacobian_reuse_tol) — same
        # fixed point and tolerance, cheaper iterations; iteration counts may then differ from pandapower's
        assert jacobian_reuse_tol >= 0.0
        self.jacobian_reuse_tol = float(jacobian_reuse_tol)
        self.pin_point_q_ranges = bool(pin_point_q_ranges)
        self.solve_opts = capi.SolveOpts(tolerance, max_iteration, (1 if self.pin_point_q_ranges else 2) if enforce_q_lims else 0, 0,
                                         int(contingency_start == 'flat'), self.jacobian_reuse_tol)
        self.np_random = np.random.default_rng(seed)

        # ---- compile the grid --------------------------------------------------
        self.case = _case_all_branches_in(net, self.act_keys, not self._topology_fixed)
        self.plan = capi.Plan(self.case, debug=self.debug)
        if self.init == 'auto':
            # (a case whose DC model is not finite — a zero-reactance branch — carries no B': 'auto' then stays flat)
            self.init = 'dc' if self.case.meta.get('calc_angles') and self.plan.info['has_dc'] else 'flat'
        if self.init == 'dc' and not self.plan.info['has_dc']:
            if not self._topology_fixed:
                raise ValueError("init='dc': this grid has no finite DC model (a zero-reactance branch has no 1/x); use "
                                 "init='auto' or 'flat'")
            self.init = 'flat'              # (a topology twin of a 'dc' parent without a DC model of its own)
        self.solve_opts.init = capi.INIT[self.init]
        self.store = ColumnStore(net)
        for tbl in ('load', 'sgen', 'storage'):
            self.store.slot(tbl, 'p_mw')
            self.store.slot(tbl, 'q_mvar')
        self.store.slot('gen', 'p_mw')
        # sampling programme first: it decides which columns are per-instance
        self._build_sampling()
        # ... and once more with the columns the STEP never reads laid out last (intermediates of the reset programme such
        # as sgen.max_p_mw of voltage_control.py:123-125, limit columns of nothing that acts): opfx_step stages the row up
        # to the last column a descriptor names (opfx_env_get_row_io), what lies behind stays in HBM
        order = sorted(self.store.ranges, key=lambda k: (k not in self._step_columns(), self.store.ranges[k][0]))
        if order != sorted(self.store.ranges, key=lambda k: self.store.ranges[k][0]):
            self.store = ColumnStore(net)
            for tbl, col in order:
                self.store.slot(tbl, col)
            self._build_sampling()
        self.n_actions = int(sum(len(idxs) for _, _, idxs in self.act_keys))
        # spaces of ONE instance (opf_env.py:124-130)
        self.observation_space = get_obs_and_state_space(net, self.obs_keys, add_time_obs, add_mean_obs,
                                                         seed=seed, bus_wise_obs=self.bus_wise_obs)
        self.state_space = get_obs_and_state_space(net, self.state_keys, seed=seed)
        self.action_space = Box(0.0, 1.0, shape=(self.n_actions,), seed=seed)
        self._env_handle = None
        self.ctx = None
        self.current_simbench_step = None
        self._reward_spec = (reward_function, reward_function_params or {})
        self.reward_function = None
        if not defer_device:
            self.attach_device()

    @classmethod
    def from_reference(cls, ref_env, batch_size=1, device='cuda:0', sampling_ops=None, **overrides):
        """Batched twin of a CONSTRUCTED reference environment (`opfgym.OpfEnv` or a subclass): the problem
        definition is read off the object — `ref_env.net`, action / observation / state keys, profiles,
        constraints, reward function with its (already estimated) scaling, the OpfEnv options, the data split,
        N-1 keys — nothing of it is restated here.  What a Python object cannot hand over as data is its
        `_sampling` override (the per-reset tail, e.g. voltage_control.py:111-133): `sampling_ops(env, ops)`
        re-expresses it as reset-kernel ops; for the reference's own classes the tails of `opfgym_amd.envs` are
        picked by class name.  Python callables in the objective / constraint seams run through the host
        fallback (opfgym_amd/host_fallback.py)."""
        from . import definition, envs as envs_mod
        defn = definition.extract(ref_env)
        names = [k.__name__ for k in type(ref_env).__mro__]
        base = MultiStageOpfEnv if 'MultiStageOpfEnv' in names else cls
        if sampling_ops is None:
            tail_cls = next((getattr(envs_mod, n) for n in names if isinstance(getattr(envs_mod, n, None), type)
                             and '_sampling_ops' in vars(getattr(envs_mod, n))), None)
            overrides_sampling = any('_sampling' in vars(k) for k in type(ref_env).__mro__ if k.__name__ not in ('OpfEnv', 'object'))
            if tail_cls is None and overrides_sampling:
                raise NotImplementedError(
                    f'{type(ref_env).__name__} overrides `_sampling`; pass sampling_ops=callable(env, ops) that '
                    f're-expresses it with the OpsBuilder operations (see opfgym_amd/envs.py for the reference classes)')
            sampling_ops = (lambda env, ops: tail_cls._sampling_ops(env, ops)) if tail_cls is not None else None
        kind = type('FromReference' + type(ref_env).__name__, (base,),
                    {'_sampling_ops': (lambda self, ops: sampling_ops(self, ops)) if sampling_ops else base._sampling_ops})
        rf = ref_env.reward_function
        if type(rf).__name__ in ('Summation', 'Replacement', 'Parameterized', 'OnlyObjective') \
                and type(rf).__module__.startswith('opfgym.'):
            prf = reward_mod.load_reward_class(type(rf).__name__)(**({} if type(rf).__name__ == 'OnlyObjective' else
                                                                  {'penalty_weight': rf.penalty_weight}))
            prf.clip_range = getattr(rf, 'clip_range', None)
            prf.scaling_params = dict(rf.scaling_params)
            for attr in ('valid_reward', 'invalid_penalty', 'invalid_objective_share'):
                if hasattr(rf, attr):
                    setattr(prf, attr, getattr(rf, attr))
        else:
            prf = rf          # a user's own reward class: evaluated on the host with the object itself (reward.SEAMS)
        cons = []
        for c_ in ref_env.constraints:
            twin = getattr(constraints_mod, type(c_).__name__, None)
            custom_values = 'get_bounded_values' in vars(c_) or 'get_boundaries' in vars(c_)
            if twin is None or custom_values or type(c_).__name__ == 'Constraint' and custom_values:
                cons.append(c_)                        # evaluated on the host through its own get_violation_metrics
                continue
            args = dict(only_worst_case_violations=c_.only_worst_case_violations, autoscale_violation=c_.autoscale_violation,
                        scale_bounded_values=c_.scale_bounded_values, penalty_factor=c_.penalty_factor,
                        penalty_power=c_.penalty_power, violation_count_penalty=c_.violation_count_penalty)
            cons.append(twin(c_.unit_type, c_.values_column, **args) if twin is constraints_mod.Constraint else twin(**args))
        objective = getattr(ref_env, 'objective_function', None)
        if getattr(objective, '__name__', '') == 'get_pandapower_costs':
            objective = None                            # the cost tables of the net (objective.py:6-32): device
        for attr in ('market_based', 'storage_efficiency'):          # constructor values the tails of envs.py read
            if hasattr(ref_env, attr):
                setattr(kind, attr, getattr(ref_env, attr))
        if type(ref_env).__name__ == 'LoadShedding':
            kind.pwl_price_columns = {'neg_price_eur_per_mw': 0, 'pos_price_eur_per_mw': 1}
        kw = dict(state_keys=defn.state_keys, profiles=defn.profiles, evaluate_on=ref_env.evaluate_on,
                  steps_per_episode=ref_env.steps_per_episode, bus_wise_obs=ref_env.bus_wise_obs,
                  reward_function=prf, diff_objective=ref_env.diff_objective,
                  add_time_obs=ref_env.add_time_obs, add_mean_obs=ref_env.add_mean_obs,
                  train_data=ref_env.train_data, test_data=ref_env.test_data, sampling_params=dict(ref_env.sampling_params),
                  custom_constraints=cons, autoscale_actions=ref_env.autoscale_actions,
                  diff_action_step_size=ref_env.diff_action_step_size,
                  clipped_action_penalty=ref_env.clipped_action_penalty, initial_action=ref_env.initial_action,
                  objective_function=objective, batch_size=batch_size, device=device, defer_device=True)
        if defn.n_minus_one_keys:
            kw.update(n_minus_one_keys=defn.n_minus_one_keys, not_converged_penalty=ref_env.not_converged_penalty)
        kw.update(overrides)
        defer = kw.pop('defer_device') if 'defer_device' in overrides else False
        kw['defer_device'] = True
        env = kind(defn.net, defn.act_keys, defn.obs_keys, **kw)        # (obs_keys already carry the add_act/add_res keys)
        env.test_steps, env.validation_steps, env.train_steps = (np.asarray(v) for v in (
            ref_env.test_steps, ref_env.validation_steps, ref_env.train_steps))
        env.definition = defn
        _notice_deviations(env)
        if not defer:
            env.attach_device()
        return env

    def attach_device(self):
        """Upload the plan, create the device evaluator and allocate the batch
        buffers.  Everything before this point is host-only problem compilation."""
        import torch
        self.torch = torch
        self.device = torch.device(self.device_spec)
        self.ctx = capi.Context(self.plan, self.device.index or 0, debug=self.debug)
        self._resolve_reward(allow_estimate=True)
        self._create_env()
        self._try_shared_slots()
        self._alloc(self.batch_size)

    def _try_shared_slots(self):
        """A grid whose instance takes more than a third of a CU's LDS runs as two wave teams of four per CU, and LDS — not
        registers or bandwidth — is what keeps a third instance out.  The plan compiler can let fill blocks that are born late
        live in the LDS slots of lower blocks that are dead by then (`opfx_debug_opts.plan_share_slots`, plan.cpp share_slots:
        wave-team kernels with full Newton only).  Where that brings the environment under a third of the LDS — three teams of
        two per CU — the environment switches to such a plan; otherwise (or with chord steps, which re-read the lower blocks)
        it keeps the one it has.  `share_lds_slots=False` switches the attempt off, and so does a `debug=` that fixes the kernel
        form (team, force_mem, kernel_v1) or the plan's slot sharing itself (plan_share_slots).  What was decided:
        `kernel_info()['shared_slots']`."""
        dbg = self.debug
        self.shared_slots = bool(self.plan.info['n_shared'])
        if self.share_lds_slots is False or self.jacobian_reuse_tol > 0.0 or self.plan.info['n_shared'] \
                or dbg.plan_share_slots or dbg.team or dbg.force_mem or dbg.kernel_v1:
            return

        def per_cu(info):
            return (160 * 1024) // (-(-info['lds_bytes_per_instance'] // 1024) * 1024)
        before = self.kernel_info()
        # (two teams of four per CU on the LDS-resident kernels: not the single-wave grids, not the grids past the LDS, whose
        #  block values are in global memory anyway)
        if before['waves_per_instance'] != 4 or per_cu(before) != 2 or self.plan.info['lds_doubles'] * 8 > 150 * 1024:
            return
        dbg = capi.debug_opts(self.debug)
        dbg.plan_share_slots = 1
        shared = capi.Plan(self.case, debug=dbg, elim_last=self.plan.elim_last)
        if not shared.info['n_shared']:
            return
        # (whether a CU then holds a third instance depends on the block storage the library picks for the new plan —
        #  two-value blocks where they buy an instance — so it is asked, not estimated: one more context and environment)
        plain = (self.plan, self.ctx)
        try:
            self.plan, self.ctx = shared, capi.Context(shared, self.device.index or 0, debug=dbg)
            self._create_env()
            gained = per_cu(self.kernel_info()) > per_cu(before)
        except capi.OpfxError:
            gained = False
        if not gained:                                         # (nothing gained: back to the plan without shared slots)
            self.plan, self.ctx = plain
            self._create_env()
        else:
            import logging
            logging.getLogger('opfgym_amd').info(
                'environment switched to a plan with shared LDS slots (%d fill blocks hosted, %d -> %d instances per CU); '
                'chord steps (jacobian_reuse_tol) are not available on it', shared.info['n_shared'], per_cu(before), per_cu(self.kernel_info()))
        self.shared_slots = bool(self.plan.info['n_shared'])

    def _resolve_reward(self, allow_estimate):
        reward_function, params = self._reward_spec                        # opf_env.py:166-175
        if not isinstance(reward_function, str):
            self.reward_function = reward_function
            return
        cls = reward_mod.load_reward_class(reward_function)
        sp = params.get('scaling_params') or {}
        needs_env = isinstance(params.get('reward_scaling'), str) and not any(
            k.startswith(('min_', 'std_')) for k in sp)
        if needs_env:
            if not allow_estimate:
                raise RuntimeError('reward scaling needs a batched estimate on the device')
            self.reward_function = reward_mod.Summation()
            self._create_env()
            self._alloc(1)
        self.reward_function = cls(env=self, **params)

    def host_definition(self):
        """Problem definition without any device object (for tools and tests)."""
        if self.reward_function is None:
            self._resolve_reward(allow_estimate=False)
        return dict(net=self.net, act_keys=self.act_keys, obs_keys=self.obs_keys,
                    profiles=self.profiles, constraints=self.constraints,
                    reward_function=self.reward_function)

    # ------------------------------------------------------------------ sampling
    def _sampling_ops(self, ops: OpsBuilder) -> None:
        """Hook for the benchmark environments' `_sampling` tails."""

    def _step_columns(self):
        """(table, column) pairs that a descriptor of the step may name: bus injections, actuators with their range and clamp
        columns, table observations, prices, per-instance voltage set-points.  Every other per-instance column is only
        written by the reset (and readable through `table_column`)."""
        hot = {(t, c) for t in ('load', 'sgen', 'storage') for c in ('p_mw', 'q_mvar')} | {('gen', 'p_mw'), ('ext_grid', 'vm_pu'), ('gen', 'vm_pu')}
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
    def _alloc(self, B):
        t, dev = self.torch, self.device
        f64 = dict(dtype=t.float64, device=dev)
        u8 = dict(dtype=t.bool, device=dev)          # one byte each; the kernel writes 0/1
        nc = max(1, self.n_device_constraints)      # (host constraints get their columns in host_fallback.finish)
        self.B = B
        self._state_valid = False
        self.x = t.zeros(B, self.nx, **f64)
        self.buf = dict(
            obs=t.zeros(B, max(1, self.n_obs_raw), **f64), reward=t.zeros(B, **f64),
            terminated=t.zeros(B, **u8), truncated=t.zeros(B, **u8), valids=t.zeros(B, nc, **u8),
            violations=t.zeros(B, nc, **f64), penalties=t.zeros(B, nc, **f64), cost=t.zeros(B, **f64),
            objective=t.zeros(B, **f64), results=t.zeros(B, self.n_results, **f64),
            mean_correction=t.zeros(B, **f64), converged=t.zeros(B, **u8),
            iterations=t.zeros(B, dtype=t.int32, device=dev), max_mismatch=t.zeros(B, **f64),
            total_iterations=t.zeros(B, dtype=t.int32, device=dev), min_pivot=t.zeros(B, **f64),
            min_pivot_bus=t.zeros(B, dtype=t.int32, device=dev))
        self.initial_obj = t.zeros(B, **f64)
        self.step_count = t.zeros(B, dtype=t.int32, device=dev)
        self.steps_dev = t.zeros(B, dtype=t.int32, device=dev)
        self._center_action = t.full((B, self.n_actions), 0.5, **f64)
        if not hasattr(self, '_gen'):
            self._gen = t.Generator(device=dev)
            self._gen.manual_seed(int(self.np_random.integers(0, 2 ** 31 - 1)))
            as_i32 = lambda a: t.as_tensor(np.asarray(a, dtype=np.int32), device=dev)
            self._pools = {'train': as_i32(self.train_steps), 'validation': as_i32(self.validation_steps),
                           'test': as_i32(self.test_steps)}

    def _io(self, action, with_initial_obj):
        io = capi.StepIO()
        io.x = self.x.data_ptr()
        io.action = action.data_ptr() if action is not None else None
        io.initial_obj = self.initial_obj.data_ptr() if with_initial_obj else None
        io.step_in_episode = self.step_count.data_ptr() if self.steps_per_episode != 1 else None
        io.outage = None
        for name, buf in self.buf.items():
            setattr(io, name, buf.data_ptr())
        return io

    def _launch_step(self, action, mode=0, with_initial_obj=False):
        # (modes that run a power flow leave results behind: opf_env.py:659 power_flow_available)
        self.power_flow_available = mode in (0, 1, 4, 5)
        self._objective_is_diff = bool(with_initial_obj)
        self._last_host = None
        if self._bb_switches and not self._topology_fixed and mode in (0, 1, 4, 5):
            return self._launch_step_by_topology(action, mode, with_initial_obj)
        io = self._io(action, with_initial_obj)
        resolve = self.on_pivot_breakdown == 'resolve' and mode in (0, 1, 4, 5)
        # (incremental actions are applied to the row in place: the rescue starts from the row as it was)
        x_before = self.x.clone() if resolve and self.diff_action_step_size and mode == 0 else None
        with self.torch.cuda.device(self.device):
            capi.check(capi.lib().opfx_step(self._env_handle, self.B, C.byref(io), C.byref(self.solve_opts),
                                            mode, capi._stream()), 'opfx_step')
        if resolve:
            self._rescue_pivot_breakdown(action, mode, with_initial_obj, x_before)

    # ---- bus-bus switches as actuators: one plan per topology -----------------------------------------------------------
    def _bb_states_after(self, action, mode):
        """[B, n] states (0 / 1) of the bus-bus switch actuators AFTER this launch has applied `action` — opf_env.py:429-481
        for those columns alone, because the topology must be known before the launch: clip, absolute or incremental
        set-point, clamp, scaling, rounding.  mode 1 applies no action: the states in the store."""
        t, d = self.torch, self._bb_act
        as_d = lambda v: t.as_tensor(np.asarray(v, dtype=np.float64), device=self.device)
        prev = self.x[:, t.as_tensor(d['slots'], device=self.device)]
        if mode == 1 or action is None:
            return t.round(prev).to(t.int64)
        a = action[:, t.as_tensor(d['cols'], device=self.device)].clamp(0.0, 1.0)          # :429
        lo, hi, sc = as_d(d['lo']), as_d(d['hi']), as_d(d['sc'])
        incremental = bool(self.diff_action_step_size) and mode == 0
        if incremental:
            sp = (a * 2 - 1) * self.diff_action_step_size * (hi - lo) + prev * sc        # :452-458
        else:
            sp = a * (hi - lo) + lo                                                      # :461
        if (not self.autoscale_actions) or incremental:                                  # :464-470
            cl, ch = as_d(d['cl']), as_d(d['ch'])
            sp = t.where(t.isnan(ch), sp, t.minimum(sp, ch))
            sp = t.where(t.isnan(cl), sp, t.maximum(sp, cl))
        return t.round(sp / sc).clamp(0, 1).to(t.int64)                                  # :472-478

    def _topology_variant(self, states):
        """The twin of this environment on the topology `states` (one 0 / 1 per bus-bus switch actuator): the same net with
        those switches set, compiled to its own case, plan and descriptor; in it the switch columns are plain columns.
        Built on first use and kept."""
        key = tuple(int(v) for v in states)
        if key not in self._topology_variants:
            net = copy.deepcopy(self.net)
            for sw, v in zip(self._bb_switches, key):
                net['switch'].at[sw['index'], 'closed'] = bool(v)
            kw = dict(self._ctor)
            kw.update(self._ctor_kwargs)
            kw.update(batch_size=1, device=self.device_spec, defer_device=False, seed=None, _topology_fixed=True,
                      reward_function=self.reward_function if self.reward_function is not None else kw['reward_function'],
                      state_keys=self.state_keys, on_pivot_breakdown='ignore', copy_outputs=False)
            action_keys, observation_keys = kw.pop('action_keys'), kw.pop('observation_keys')
            var = type(self).__new__(type(self))                  # (the same class: its `_sampling_ops` decides the row layout)
            var.__dict__.update(self._pre_init_attrs)
            BatchedOpfEnv.__init__(var, net, action_keys, observation_keys, **kw)
            if var.store.n != self.store.n or var.n_obs_raw != self.n_obs_raw or var.n_constraints != self.n_constraints:
                raise RuntimeError('a topology variant laid its rows out differently from its parent')
            var._results_from = self._results_map(var)
            self._topology_variants[key] = var
        return self._topology_variants[key]

    def _results_map(self, var):
        """(positions in this environment's result bank, positions in the variant's) of the same physical quantities:
        bus voltages and angles through the net's bus numbers (a fused bus serves both of its net buses), branch
        loadings through the net elements, slack powers through the ext_grids; derived rows by position."""
        c, v = self.case, var.case
        nb, nbv = c.nb, v.nb
        mine, theirs = [], []
        inv = {}
        for net_bus, i in c.bus_lookup.items():
            inv.setdefault(i, net_bus)
        for i in range(nb):
            j = v.bus_lookup.get(inv[i], -1) if i in inv else -1
            if j >= 0:
                for off_m, off_v in ((0, 0), (nb, nbv)):
                    mine.append(off_m + i); theirs.append(off_v + j)
        vbr = {(int(k), int(e)): n for n, (k, e) in enumerate(zip(v.br_kind, v.br_elem))}
        for n, (k, e) in enumerate(zip(c.br_kind, c.br_elem)):
            if (int(k), int(e)) in vbr:
                mine.append(2 * nb + n); theirs.append(2 * nbv + vbr[(int(k), int(e))])
        ref_m, ref_v = np.flatnonzero(c.bus_type == REF), np.flatnonzero(v.bus_type == REF)
        for r, i in enumerate(ref_m):
            j = v.bus_lookup.get(inv.get(int(i), -1), -1)
            hit = np.flatnonzero(ref_v == j)
            if len(hit):
                for q in range(2):
                    mine.append(2 * nb + c.nbr + q * len(ref_m) + r); theirs.append(2 * nbv + v.nbr + q * len(ref_v) + int(hit[0]))
        base_m, base_v = 2 * nb + c.nbr + 2 * len(ref_m), 2 * nbv + v.nbr + 2 * len(ref_v)
        for i in range(nb):                                               # (reactive power of the generators per bus)
            j = v.bus_lookup.get(inv[i], -1) if i in inv else -1
            if j >= 0:
                mine.append(base_m + i); theirs.append(base_v + j)
        for k in range(min(len(self._xres), len(var._xres))):             # (derived rows: allocated in the same order)
            mine.append(base_m + nb + k); theirs.append(base_v + nbv + k)
        t = self.torch
        return (t.as_tensor(mine, dtype=t.int64, device=self.device), t.as_tensor(theirs, dtype=t.int64, device=self.device))

    def _launch_step_by_topology(self, action, mode, with_initial_obj):
        """One step for a batch whose instances sit on different topologies (bus-bus switch actuators): the rows are
        grouped by the switch states the action leaves them in, every group is stepped by the twin compiled for that
        topology (its rows gathered into a compact batch, as the pivot rescue does) and scattered back.  One host
        synchronisation per step (which topologies occur) and one launch per topology that occurs."""
        t, b = self.torch, self.buf
        states = self._bb_states_after(action, mode)
        weights = t.as_tensor([1 << k for k in range(states.shape[1])], dtype=t.int64, device=self.device)
        codes = (states * weights).sum(dim=1)
        for code in t.unique(codes).cpu().tolist():
            idx = (codes == code).nonzero().flatten()
            var = self._topology_variant([(code >> k) & 1 for k in range(states.shape[1])])
            n = int(idx.numel())
            x2 = self.x[idx].contiguous()
            tmp = {name: buf[idx].contiguous() for name, buf in b.items() if name != 'results'}
            tmp['results'] = t.full((n, var.n_results), float('nan'), dtype=t.float64, device=self.device)
            io = capi.StepIO()
            io.x = x2.data_ptr()
            act2 = action[idx].contiguous() if action is not None else None
            io.action = act2.data_ptr() if act2 is not None else None
            init2 = self.initial_obj[idx].contiguous() if with_initial_obj else None
            io.initial_obj = init2.data_ptr() if init2 is not None else None
            cnt2 = self.step_count[idx].contiguous() if self.steps_per_episode != 1 else None
            io.step_in_episode = cnt2.data_ptr() if cnt2 is not None else None
            io.outage = None
            for name, buf in tmp.items():
                setattr(io, name, buf.data_ptr())
            with t.cuda.device(self.device):
                # (the twin's own start: its topology decides whether it has a DC model; everything else is the parent's)
                capi.check(capi.lib().opfx_step(var._env_handle, n, C.byref(io), C.byref(var.solve_opts), mode,
                                                capi._stream()), 'opfx_step (topology variant)')
            self.x[idx] = x2
            for name, buf in tmp.items():
                if name != 'results':
                    b[name][idx] = buf
            mine, theirs = var._results_from
            rows = t.full((n, self.n_results), float('nan'), dtype=t.float64, device=self.device)
            rows[:, mine] = tmp['results'][:, theirs]
            b['results'][idx] = rows

    def contingency_results(self, branch):
        """Result bank [B, n_results] and convergence flags [B] of the CURRENT set-points with case branch `branch` out of
        service — one power flow per instance, no action applied, nothing of the step's outputs touched (the host
        fallback's view of one N-1 contingency, security_constrained.py:50-56)."""
        t = self.torch
        if self._env_handle_base_only is None:
            raise RuntimeError('contingency_results: only for environments with host callables and N-1 keys')
        io = capi.StepIO()
        io.x = self.x.data_ptr()
        io.step_in_episode = self.step_count.data_ptr() if self.steps_per_episode != 1 else None
        out = t.full((self.B,), int(branch), dtype=t.int32, device=self.device)
        res = t.empty(self.B, self.n_results, dtype=t.float64, device=self.device)
        conv = t.zeros(self.B, dtype=t.uint8, device=self.device)
        io.outage, io.results, io.converged = out.data_ptr(), res.data_ptr(), conv.data_ptr()
        with t.cuda.device(self.device):
            capi.check(capi.lib().opfx_step(self._env_handle_base_only, self.B, C.byref(io), C.byref(self.solve_opts),
                                            1, capi._stream()), 'opfx_step (one contingency)')
        return res, conv

    PIVOT_BREAKDOWN = 1e-8

    def _rescue_pivot_breakdown(self, action, mode, with_initial_obj, x_before):
        """Rows whose factorisation broke down (not converged, min_pivot < 1e-8) once more, on a plan that eliminates the
        buses named by `min_pivot_bus` last (see `on_pivot_breakdown`): gathered into a compact batch, stepped by a second
        environment object on that plan (same descriptor), scattered back.  GPU only — there is no CPU fallback."""
        t, b = self.torch, self.buf
        bad = (b['converged'] == 0) & (b['min_pivot'] < self.PIVOT_BREAKDOWN) & (b['min_pivot_bus'] >= 0)
        if not bool(bad.any()):                                            # (the one host synchronisation of this option)
            return
        idx = bad.nonzero().flatten()
        buses = tuple(sorted(set(b['min_pivot_bus'][idx].cpu().tolist())))
        if buses not in self._rescue_envs:
            self._drop_rescue_envs(keep=self.MAX_RESCUE_PLANS - 1)         # (each holds a plan, a context and a device copy)
            plan = capi.Plan(self.case, elim_last=buses, debug=self.debug)
            ctx = capi.Context(plan, self.device.index or 0, debug=self.debug)
            h = C.c_void_p()
            capi.check(capi.lib().opfx_env_create(ctx.handle, C.byref(self._env_desc), C.byref(h)), 'opfx_env_create (rescue plan)')
            self._rescue_envs[buses] = (plan, ctx, h)
        self._rescue_envs[buses] = self._rescue_envs.pop(buses)            # most recently used last
        _, _, handle = self._rescue_envs[buses]
        n = int(idx.numel())
        x2 = (x_before if x_before is not None else self.x)[idx].contiguous()
        tmp = {name: buf[idx].contiguous() for name, buf in b.items()}
        io = capi.StepIO()
        io.x = x2.data_ptr()
        act2 = action[idx].contiguous() if action is not None else None
        io.action = act2.data_ptr() if act2 is not None else None
        init2 = self.initial_obj[idx].contiguous() if with_initial_obj else None
        io.initial_obj = init2.data_ptr() if init2 is not None else None
        cnt2 = self.step_count[idx].contiguous() if self.steps_per_episode != 1 else None
        io.step_in_episode = cnt2.data_ptr() if cnt2 is not None else None
        io.outage = None
        for name, buf in tmp.items():
            setattr(io, name, buf.data_ptr())
        with t.cuda.device(self.device):
            capi.check(capi.lib().opfx_step(handle, n, C.byref(io), C.byref(self.solve_opts), mode, capi._stream()), 'opfx_step (rescue plan)')
        self.x[idx] = x2
        for name, buf in tmp.items():
            b[name][idx] = buf
        self.pivot_rescues += n
        self.pivot_rescues_recovered += int(tmp['converged'].sum().item())

    def _as_action(self, action):
        t = self.torch
        if t.is_tensor(action) and action.dtype == t.float64 and action.device == self.device \
                and action.is_contiguous() and action.shape == (self.B, self.n_actions):
            return action
        if not t.is_tensor(action):
            action = t.as_tensor(np.asarray(action, dtype=np.float64))
        action = action.to(device=self.device, dtype=t.float64).reshape(self.B, self.n_actions).contiguous()
        return action

    # ------------------------------------------------------------------ gymnasium-shaped API
    def reset(self, seed=None, options=None):
        """opf_env.py:177-220 for the whole batch.  (The returned observation is a view of a persistent output
        buffer, see `step`.)  options: 'test' (bool),
        'step' (int or [B] array), plus 'noise' [B,n_noise] / 'uniform'
        [B,n_uniform] / 'initial_action' [B,na] to replay explicit draws.

        When the observation needs a power flow (`pf_for_obs`) and it fails for some instances, those
        instances are sampled again with fresh random draws until they converge — the reference's
        `return self.reset()` (opf_env.py:211-214), per row: the others keep their state."""
        t = self.torch
        if seed is not None:
            self.np_random = np.random.default_rng(seed)
            self._gen.manual_seed(int(seed))
        options = options or {}
        self.power_flow_available = False                                  # opf_env.py:181 (a power flow of this reset sets it again)
        self._sample_and_initialise(options)
        if self.pf_for_obs and self.resample_failed_resets and not bool(self.buf['converged'].all()):
            ok = self.buf['converged'].clone()
            state = lambda: [self.x, self.steps_dev, self.initial_obj] + \
                ([self.sampling_mode] if self.per_source else []) + list(self.buf.values())
            for _ in range(self.max_reset_retries):
                saved = [v.clone() for v in state()]
                self._sample_and_initialise({'test': self.test})           # fresh steps and draws for every row
                again = ~ok                                                # rows that take the new sample
                for cur, old in zip(state(), saved):
                    sel = again.view(-1, *([1] * (cur.dim() - 1)))
                    cur.copy_(t.where(sel, cur, old))
                ok = ok | self.buf['converged']                            # (kept rows carry their old flag = True)
                self.current_simbench_step = None
                if bool(ok.all()):
                    break
            else:
                raise RuntimeError(f'power flow failed in reset for some instances after '
                                   f'{self.max_reset_retries} new samples (opf_env.py:211-214)')
        obs = self._finish_obs()
        return (obs.clone() if self.copy_outputs else obs), {}

    max_reset_retries = 20

    def _sample_and_initialise(self, options):
        """One pass of opf_env.py:196-216 for every row: sample the state, apply the initial action, run
        the power flow when the observation needs it (convergence flags in buf['converged'])."""
        t = self.torch
        B = self.B
        dev = self.device
        self.test = bool(options.get('test', False))
        step = options.get('step', None)
        in_kernel_pool = None
        if step is None:                                                   # opf_env.py:327-333
            key = 'test' if (self.test and self.evaluate_on == 'test') else \
                ('validation' if (self.test and self.evaluate_on == 'validation') else 'train')
            if self.uses_profiles:
                # the reset kernel draws the step itself (a counter-based generator keyed by a per-reset seed and
                # the instance number): no launch of its own for the draw, no host round trip
                in_kernel_pool = self._pools[key]
            else:
                self.steps_dev.zero_()
            self.current_simbench_step = None
        else:
            step = np.broadcast_to(np.asarray(step, dtype=np.int64), (B,))
            if self.uses_profiles:                                         # (without time series the step is not used)
                n_steps = len(self.profiles[('load', 'q_mvar')])
                if (step < 0).any() or (step >= n_steps).any():            # :335 (the reference's .loc raises KeyError)
                    raise KeyError(f"reset option 'step' outside the profile horizon [0, {n_steps})")
            else:
                step = np.zeros(B, dtype=np.int64)
            self.current_simbench_step = step.copy()
            self.steps_dev.copy_(t.as_tensor(step.astype(np.int32)))

        def as_dev(a):
            if t.is_tensor(a):
                return a.to(device=dev, dtype=t.float64).contiguous()
            return t.as_tensor(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
        data_distr = self.test_data if self.test else self.train_data
        noise_t, interp_t, nnf = self._profile_draws(data_distr, options, as_dev)
        # uniform / normal draws of the `_sampling` programme: explicit ones replay; otherwise the reset kernel makes
        # them itself (counter-based, keyed by the per-reset seed below and the instance number) — no launch for them
        uni_t = options.get('uniform')
        uni_t = as_dev(uni_t) if uni_t is not None else None
        nrm_t = options.get('normal')
        nrm_t = as_dev(nrm_t) if nrm_t is not None else None
        rio = capi.ResetIO()
        rio.step_idx = self.steps_dev.data_ptr()
        rio.rng_seed = int(self.np_random.integers(1, 2 ** 63 - 1))
        if in_kernel_pool is not None:
            rio.step_pool, rio.n_step_pool = in_kernel_pool.data_ptr(), len(in_kernel_pool)
            rio.step_out = self.steps_dev.data_ptr()
        rio.noise = noise_t.data_ptr() if noise_t is not None else None
        rio.interp = interp_t.data_ptr() if interp_t is not None else None
        rio.uniform = uni_t.data_ptr() if uni_t is not None else None
        rio.normal = nrm_t.data_ptr() if nrm_t is not None else None
        rio.normal_noise_factor = nnf
        rio.x = self.x.data_ptr()
        rio.keep_state = 1 if (self.carry_over_state and self._state_valid) else 0
        self._state_valid = True
        mode_t = None
        if self.mixed and data_distr == 'mixed':                           # opf_env.py:242-251
            mode_t = options.get('mode')
            if mode_t is None:
                r = t.rand(B, generator=self._gen, device=dev, dtype=t.float64)
                p0, p1 = self.data_probabilities[0], self.data_probabilities[1]
                mode_t = ((r >= p0).to(t.int32) + (r >= p1).to(t.int32))
            else:
                mode_np = np.broadcast_to(np.asarray(mode_t, dtype=np.int32), (B,)).copy()
                if (mode_np < 0).any() or (mode_np > 2).any():
                    raise ValueError("reset option 'mode': 0 (profile row), 1 (uniform), 2 (normal around the mean)")
                mode_t = t.as_tensor(mode_np).to(dev)
            rio.mode = mode_t.contiguous().data_ptr()
            self.sampling_mode = mode_t
        elif self.per_source:                                              # train and test distribution differ
            mode_t = t.full((B,), self.source_of[data_distr], dtype=t.int32, device=dev)
            rio.mode = mode_t.data_ptr()
            self.sampling_mode = mode_t
        act = options.get('initial_action')
        if act is None:
            if self.initial_action == 'random':                            # :201-203
                act = t.rand(B, self.n_actions, generator=self._gen, device=dev, dtype=t.float64)
            else:
                act = self._center_action                                  # :206
        act = self._as_action(act)
        if not self.pf_for_obs:
            # no power flow needed for the observation: sampling, initial action and the table
            # observation in ONE launch (opf_env.py:199-207, 217-218)
            rio.action = act.data_ptr()
            rio.obs = self.buf['obs'].data_ptr()
        with t.cuda.device(self.device):
            capi.check(capi.lib().opfx_reset(self._env_handle, B, C.byref(rio), capi._stream()), 'opfx_reset')
        if self.steps_per_episode != 1:                                    # (single-step episodes never read the counter:
            self.step_count.zero_()                                        #  no launch for it in the reset + step cycle)
        if self.pf_for_obs:                                                # :209-216
            self._launch_step(act, mode=4, with_initial_obj=False)
            if self._host_finisher is not None:
                self._last_host = self._host_finisher.finish(4)
            self.initial_obj.copy_(self.buf['objective'])

    def _profile_draws(self, data_distr, options, as_dev):
        """Noise and interpolation draws of `_set_simbench_state` (opf_env.py:345-360) for every row:
        (noise tensor or None, interpolation tensor or None, normal-noise factor).  Explicit draws in `options`
        ('noise', 'interp') are replayed instead."""
        t, B, dev = self.torch, self.B, self.device
        noisy = self.n_noise and (data_distr == 'noisy_simbench' or 'noise_factor' in self.sampling_params
                                  or (self.mixed and data_distr == 'mixed'))
        nf = self.noise_factor if noisy else 0.0
        normal_noise = noisy and self.noise_distribution == 'normal'
        noise_t = options.get('noise')
        if noise_t is not None:
            noise_t = as_dev(noise_t)          # uniform mode: factors; normal mode: standard-normal draws
        elif noisy and normal_noise:
            noise_t = t.randn(B, self.n_noise, generator=self._gen, device=dev, dtype=t.float64)   # :359-360
        elif noisy and nf:
            noise_t = t.rand(B, self.n_noise, generator=self._gen, device=dev, dtype=t.float64) * (nf * 2) \
                + (1 - nf)                                                     # :354-355
        interp_t = options.get('interp')
        if interp_t is not None:
            interp_t = as_dev(interp_t)
        elif self.interpolate_steps and self.uses_profiles:
            interp_t = t.rand(B, len(self.tables), generator=self._gen, device=dev, dtype=t.float64)   # :348
        return noise_t, interp_t, (float(nf) if normal_noise else 0.0)

    def step(self, action):
        """opf_env.py:374-419 for the whole batch: (obs, reward, terminated,
        truncated, info) as torch tensors on the device.

        LIFETIME: the returned tensors are the environment's persistent output buffers (or views of them) —
        the next `reset()` / `step()` overwrites them in place.  A rollout loop that keeps them across calls
        must `.clone()` them, or construct the environment with `copy_outputs=True`."""
        t = self.torch
        action = self._as_action(action)
        if self.validate_actions:
            assert not bool(t.isnan(action).any())                         # :382
        if self.steps_per_episode != 1:
            self.step_count += 1
        self._launch_step(action, mode=0, with_initial_obj=self.diff_objective)
        b = self.buf
        host = self._host_finisher.finish(0, self.initial_obj if self.diff_objective else None) \
            if self._host_finisher is not None else None
        self._last_host = host
        info = {'valids': b['valids'] if host is None else host['valids'],
                'violations': b['violations'] if host is None else host['violations'],
                'unscaled_penalties': b['penalties'] if host is None else host['penalties'], 'cost': b['cost'],
                'converged': b['converged'], 'iterations': b['iterations'],
                'max_mismatch': b['max_mismatch'], 'objective': b['objective'],
                'total_iterations': b['total_iterations'], 'min_pivot': b['min_pivot'], 'min_pivot_bus': b['min_pivot_bus']}
        out = (self._finish_obs(), b['reward'], b['terminated'], b['truncated'], info)
        if self.copy_outputs:
            out = tuple(v.clone() for v in out[:4]) + ({k: (v.clone() if self.torch.is_tensor(v) else v) for k, v in info.items()},)
        return out

    def _finish_obs(self):
        """add_mean_obs / add_time_obs post-processing (opf_env.py:539-547)."""
        t = self.torch
        obs = self.buf['obs'][:, :self.n_obs_raw]
        if self.bus_wise_obs or self.add_mean_obs:
            segs, off = [], 0
            for (unit, col, idxs), n in zip(self.obs_keys, self.obs_segments):
                seg = obs[:, off:off + n]
                off += n
                if self.bus_wise_obs and unit == 'load':                   # opf_env.py:535-536, 806-810
                    buses = self.net.load.iloc[np.asarray(idxs)].bus.to_numpy()
                    uniq, inv = np.unique(buses, return_inverse=True)
                    agg = t.zeros(seg.shape[0], len(uniq), dtype=seg.dtype, device=seg.device)
                    agg.index_add_(1, t.as_tensor(inv, device=seg.device), seg)
                    seg = agg
                segs.append(seg)
            parts = list(segs)
            if self.add_mean_obs:                                          # :539-542
                parts += [sg.mean(dim=1, keepdim=True) for sg in segs if sg.shape[1] > 1]
        else:
            parts = [obs]
        if self.add_time_obs:
            # always the step the instance is at NOW (steps_dev is what the reset kernel sampled at and what
            # multi-stage episodes / partial resets advance; a host copy of the reset option would go stale)
            tobs = get_simbench_time_observation(self.steps_dev.cpu().numpy())   # intended semantics (defect D1)
            parts = [t.as_tensor(tobs, dtype=t.float64, device=self.device)] + parts
        return t.cat(parts, dim=1) if len(parts) > 1 else parts[0]

    # ------------------------------------------------------------------ state / validity / objective
    def _key_values(self, unit, col, idxs):
        """`net[unit].loc[idxs, col]` for every instance, [B, len(idxs)]: a result column, a per-instance table
        column of x, or a column that is the same for all instances."""
        t = self.torch
        idxs = np.asarray(list(idxs))
        if unit.startswith('res_'):
            rows = self.store.rows(unit[4:], idxs)
            return self.result_table(unit[4:], col)[:, t.as_tensor(np.asarray(rows), device=self.device)]
        rows = t.as_tensor(np.asarray(self.store.rows(unit, idxs)), device=self.device)
        if (unit, col) in self.store.ranges:
            return self.table_column(unit, col)[:, rows]
        const = t.as_tensor(self.net[unit][col].to_numpy(dtype=float), dtype=t.float64, device=self.device)
        return const[rows].expand(self.B, -1)

    def get_state(self):
        """opf_env.py:551-556 for the batch: the values behind `state_keys`, [B, n_state] — the full state of a
        partially observable environment (`examples/partial_obs.py`)."""
        t = self.torch
        parts = []
        for unit, col, idxs in self.state_keys:
            seg = self._key_values(unit, col, idxs)
            if self.bus_wise_obs and unit == 'load':                       # opf_env.py:535-536, 806-810
                buses = self.net.load.iloc[np.asarray(list(idxs))].bus.to_numpy()
                uniq, inv = np.unique(buses, return_inverse=True)
                agg = t.zeros(seg.shape[0], len(uniq), dtype=seg.dtype, device=seg.device)
                agg.index_add_(1, t.as_tensor(inv, device=seg.device), seg)
                seg = agg
            parts.append(seg)
        return t.cat(parts, dim=1) if parts else t.zeros(self.B, 0, dtype=t.float64, device=self.device)

    def run_power_flow(self):
        """opf_env.py:646-662 for the batch: evaluate the CURRENT set-points (no action is applied) — power flow,
        objective, violations.  Returns the converged flags, [B]."""
        self._launch_step(self._center_action, mode=1)
        if self._host_finisher is not None:
            self._last_host = self._host_finisher.finish(1)
        return self.buf['converged']

    def ensure_power_flow_available(self):
        if not self.power_flow_available:                                  # opf_env.py:682-684
            raise PowerFlowNotAvailable('Please call `run_power_flow` first!')

    def set_power_flow_unavailable(self):
        self.power_flow_available = False                                  # opf_env.py:690-694

    def is_state_valid(self):
        """opf_env.py:613-618 for the batch: no constraint violated (and the power flow converged), [B] bool."""
        self.ensure_power_flow_available()
        # (no constraints at all: an empty all() is True, as in the reference)
        valids = self._last_host['valids'][:, :self.n_constraints] if self._last_host is not None \
            else self.buf['valids'][:, :self.n_device_constraints]
        return valids.all(dim=1) & self.buf['converged']

    def get_objective(self):
        """opf_env.py:635-638 for the batch: the current value of the objective function, [B] (never the
        difference to the initial objective, whatever `diff_objective` says)."""
        self.ensure_power_flow_available()
        obj = self.buf['objective']
        return obj + self.initial_obj if self._objective_is_diff else obj

    # ------------------------------------------------------------------ helpers
    def get_current_actions(self, from_results_table=True):
        """opf_env.py:566-588 for the batch: (set-point·scaling − min)/(max − min) per action,
        [B, n_actions].  `res_<unit>.<col>` equals set-point·scaling for a unit that takes part in the power flow and
        zero for one that does not (out of service, bus outside the compiled case), so both variants read the same
        columns of x; the table variant (`from_results_table=False`) has no such mask (opf_env.py:577-578)."""
        t = self.torch
        d = self._act_desc
        x = self.x
        sp = x[:, d['slot']] * d['scaling']
        if from_results_table:
            sp = sp * d['part']
        lo = t.where(d['lo_slot'] >= 0, x[:, d['lo_slot'].clamp(min=0)], d['lo_const'])
        hi = t.where(d['hi_slot'] >= 0, x[:, d['hi_slot'].clamp(min=0)], d['hi_const'])
        return (sp - lo) / (hi - lo)

    def get_actions(self):
        """opf_env.py:590-600."""
        return self.get_current_actions()

    def results(self):
        """Result bank of the last step as a dict of tensors (net.res_* columns)."""
        c = self.case
        nb, nbr = c.nb, c.nbr
        nref = int((c.bus_type == REF).sum())
        r = self.buf['results']
        return dict(vm_pu=r[:, :nb], va_degree=r[:, nb:2 * nb], loading_percent=r[:, 2 * nb:2 * nb + nbr],
                    p_ext_mw=r[:, 2 * nb + nbr:2 * nb + nbr + nref],
                    q_ext_mvar=r[:, 2 * nb + nbr + nref:2 * nb + nbr + 2 * nref],
                    q_gen_mvar=r[:, 2 * nb + nbr + 2 * nref:])

    def result_table(self, unit, col):
        """`net.res_<unit>.<col>` of the last step for all instances, [B, n_rows]
        in the row order of `net.<unit>` (NaN for de-energised elements)."""
        t = self.torch
        if unit == 'ext_grid' or (unit == 'gen' and col == 'q_mvar'):
            return self._generator_result_table(unit, col)
        idx = self._result_index(unit, col, self.net[unit].index)
        gather = t.as_tensor(np.where(idx < 0, 0, idx), device=self.device)
        out = self.buf['results'][:, gather]
        if (idx < 0).any():
            out = out.clone()
            out[:, t.as_tensor(idx < 0, device=self.device)] = float('nan')
        return out

    def _generator_result_table(self, unit, col):
        """`res_ext_grid.p_mw / q_mvar` and `res_gen.q_mvar` from the solver's per-BUS entries of the result bank and the
        shares of `case.generator_dispatch` (the same affine map the kernel applies in its derived rows, which exist only
        where the compiled environment reads them): NaN for an ext_grid outside the power flow, zero for such a generator
        or one whose bus is de-energised in the instance."""
        t = self.torch
        c = self.case
        nb, nbr = c.nb, c.nbr
        ref_buses = np.flatnonzero(c.bus_type == REF)
        nref = len(ref_buses)
        ordinal = {int(b): k for k, b in enumerate(ref_buses)}
        share = self._generator_shares()[unit]
        n = len(self.net[unit])
        src, a, b, vm_of = np.zeros(n, np.int64), np.zeros(n), np.zeros(n), np.zeros(n, np.int64)
        for pos in range(n):
            i = int(share['bus'][pos])
            if i < 0:
                a[pos] = np.nan if unit == 'ext_grid' else 0.0
                continue
            vm_of[pos] = i
            if unit == 'ext_grid' and col == 'p_mw':
                src[pos], a[pos], b[pos] = 2 * nb + nbr + ordinal[i], 0.0, share['p_b'][pos]
            else:
                src[pos] = 2 * nb + nbr + nref + ordinal[i] if c.bus_type[i] == REF else 2 * nb + nbr + 2 * nref + i
                a[pos], b[pos] = share['q_a'][pos], share['q_b'][pos]
        res = self.buf['results']
        dev = lambda v: t.as_tensor(v, device=self.device)
        out = dev(a) + dev(b) * res[:, dev(src)]
        dead = t.isnan(res[:, dev(vm_of)]) & dev(share['bus'] >= 0)          # (|V| of a de-energised bus is NaN)
        return t.where(dead, t.zeros_like(out), out) if unit == 'gen' else out

    def table_column(self, unit, col):
        """Current per-instance values of a table column held in x, [B, n_rows]."""
        off, n = self.store.ranges[(unit, col)]
        return self.x[:, off:off + n]

    def sample_objective_penalty(self, num_samples, draws=None):
        """One batched reset + random action + power flow (reward.py:181-196): Σobjective and Σpenalty per
        sample (NaN where the power flow failed).  `draws` replays explicit inputs instead of drawing them:
        dict(step=[n], uniform=[n, n_uniform] | None, noise=[n, n_noise] raw U[0,1) draws | None,
        action=[n, n_actions])."""
        old = (self.B, self.x, self.buf, self.initial_obj, self.step_count, self.steps_dev, self._center_action)
        self._alloc(int(num_samples))
        t = self.torch
        try:
            if draws is None:
                self.reset()
                action = t.rand(self.B, self.n_actions, generator=self._gen, device=self.device, dtype=t.float64)
            else:
                opts = {'step': np.asarray(draws['step'])}
                if draws.get('uniform') is not None and self.n_uniform:
                    opts['uniform'] = draws['uniform']
                if draws.get('noise') is not None and self.n_noise and np.size(draws['noise']):
                    nf = self.noise_factor
                    raw = np.asarray(draws['noise'], dtype=float)
                    opts['noise'] = raw if self.noise_distribution == 'normal' else raw * nf * 2 + (1 - nf)   # opf_env.py:354-355
                self.reset(options=opts)
                action = self._as_action(draws['action'])
            self.step_count += 1
            # absolute set-points as `_apply_actions(action)` without a step size, contingencies included
            self._launch_step(action, mode=5)
            pen_t = self.buf['penalties'][:, :self.n_device_constraints]
            if self._host_finisher is not None:
                pen_t = self._host_finisher.finish(5)['penalties'][:, :self.n_constraints]
            conv = self.buf['converged'].cpu().numpy()
            obj = self.buf['objective'].cpu().numpy().copy()
            pen = pen_t.sum(dim=1).cpu().numpy().copy()
            obj[~conv] = np.nan
            pen[~conv] = np.nan
        finally:
            self.B, self.x, self.buf, self.initial_obj, self.step_count, self.steps_dev, self._center_action = old
        return obj, pen

    def kernel_info(self) -> dict:
        """Launch configuration of the fused step kernel (report / diagnostics): wavefronts per instance, LDS bytes per
        instance, instances resident per CU (0 before the first step) and whether the two-value block storage is used."""
        team, lds, per_cu, nblk, nfour, spec = C.c_int32(), C.c_int64(), C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        capi.check(capi.lib().opfx_env_get_info(self._env_handle, C.byref(team), C.byref(lds), C.byref(per_cu)), 'opfx_env_get_info')
        capi.check(capi.lib().opfx_env_get_storage(self._env_handle, C.byref(nblk), C.byref(nfour)), 'opfx_env_get_storage')
        capi.check(capi.lib().opfx_env_get_spec(self._env_handle, C.byref(spec)), 'opfx_env_get_spec')
        return dict(waves_per_instance=team.value, lds_bytes_per_instance=lds.value, instances_per_cu=per_cu.value,
                    packed=nfour.value < nblk.value, n_blk=nblk.value, n_four_value=nfour.value,
                    spec=spec.value, shared_slots=bool(self.plan.info['n_shared']), debug=self.debug.as_dict(),
                    x_columns=self.store.n, x_columns_read=self._x_columns_read())          # (SPEC bits of the plain step kernel: 1 no PV bus, 2 no modifiers)

    def _x_columns_read(self):
        n = C.c_int32()
        capi.check(capi.lib().opfx_env_get_row_io(self._env_handle, C.byref(n)), 'opfx_env_get_row_io')
        return n.value

    MAX_RESCUE_PLANS = 8

    def _drop_rescue_envs(self, keep=0):
        """Destroy rescue environments (oldest first) until `keep` are left: the env handle before its context goes."""
        cache = getattr(self, '_rescue_envs', None) or {}
        while len(cache) > keep:
            _, _, h = cache.pop(next(iter(cache)))
            capi.lib().opfx_env_destroy(h)
        self._rescue_envs = cache

    def close(self):
        self._drop_rescue_envs()
        for var in getattr(self, '_topology_variants', {}).values():
            var.close()
        self._topology_variants = {}
        if getattr(self, '_env_handle_base_only', None) is not None:
            capi.lib().opfx_env_destroy(self._env_handle_base_only)
            self._env_handle_base_only = None
        if self._env_handle is not None:
            capi.lib().opfx_env_destroy(self._env_handle)
            self._env_handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SecurityConstrainedOpfEnv(BatchedOpfEnv):
    """security_constrained.py:7-35: same arguments; the K contingency solves of
    every instance run inside the same kernel launch as its base case."""

    def __init__(self, *args, n_minus_one_keys, not_converged_penalty=1, **kwargs):
        super().__init__(*args, n_minus_one_keys=n_minus_one_keys,
                         not_converged_penalty=not_converged_penalty, **kwargs)


class MultiStageOpfEnv(BatchedOpfEnv):
    """multi_stage.py:4-58 for the batch: after every step the instances that continue their
    episode move on to the next SimBench time step (state re-sampled by the reset kernel at
    step+1, `_sampling` tail included); crossing the train/test split truncates the episode."""

    def __init__(self, *args, steps_per_episode: int = 4, **kwargs):
        assert steps_per_episode > 1, 'At least two steps required for a multi-stage OPF.'
        super().__init__(*args, steps_per_episode=steps_per_episode, **kwargs)

    def attach_device(self):
        super().attach_device()
        t = self.torch
        horizon = len(self.profiles[('load', 'q_mvar')])
        kind = np.zeros(horizon + 1, dtype=np.int8)            # 0 train, 1 validation, 2 test
        kind[np.asarray(self.validation_steps, dtype=int)] = 1
        kind[np.asarray(self.test_steps, dtype=int)] = 2
        self._step_kind = t.as_tensor(kind, device=self.device)
        self._x_next = t.zeros_like(self.x)

    def step(self, action):
        t = self.torch
        obs, reward, terminated, truncated, info = super().step(action)
        new_step = (self.steps_dev + 1).long()
        kind = self._step_kind[new_step.clamp(max=len(self._step_kind) - 1)]
        crossing = (kind == 0) if self.test else (kind != 0)             # multi_stage.py:32-39
        truncated = truncated | crossing
        terminated = terminated | (self.step_count >= self.steps_per_episode)   # :42-43
        cont = ~(terminated | truncated)
        if bool(cont.any()):
            # re-sample every row at step+1 into a scratch store, keep it for the continuing rows
            obs = obs.clone()          # (the observation buffer is about to be overwritten)
            if self.pf_for_obs:        # ... and so are the step's outputs, by the power flow of the new state
                reward = reward.clone()
                info = {k: (v.clone() if t.is_tensor(v) else v) for k, v in info.items()}
            steps_old, x_old = self.steps_dev, self.x
            self.steps_dev = (self.steps_dev + 1).clamp(max=len(self._step_kind) - 2).int()
            self._x_next.copy_(x_old)      # (only the sampled columns change: multi_stage.py:49-56 works on the same net)
            self.x = self._x_next
            self._resample_current()
            if self.pf_for_obs:        # multi_stage.py:52-53: power flow of the new state with the set-points kept
                self._launch_step(self._center_action, mode=1)
            new_obs = self._finish_obs()
            self.x = t.where(cont[:, None], self._x_next, x_old)
            self._x_next = x_old
            self.steps_dev = t.where(cont, self.steps_dev, steps_old)
            obs = t.where(cont[:, None], new_obs, obs)
        return obs, reward, terminated, truncated, info

    def _resample_current(self):
        """reset kernel at self.steps_dev into self.x together with the table observation, no action
        (what `_sampling(step=new_step)` + `_get_obs` do at multi_stage.py:49-56) — one launch.  `_sampling`
        merges the environment's sampling_params (opf_env.py:228-237), so every stage draws fresh noise /
        interpolation weights exactly as a reset does."""
        B, t = self.B, self.torch
        rio = capi.ResetIO()
        rio.step_idx = self.steps_dev.data_ptr()
        data_distr = self.test_data if self.test else self.train_data
        noise_t, interp_t, nnf = self._profile_draws(
            data_distr, {}, lambda a: a.to(device=self.device, dtype=t.float64).contiguous())
        rio.noise = noise_t.data_ptr() if noise_t is not None else None
        rio.interp = interp_t.data_ptr() if interp_t is not None else None
        rio.normal_noise_factor = nnf
        uni = t.rand(B, self.n_uniform, generator=self._gen, device=self.device, dtype=t.float64) \
            if self.n_uniform else None
        rio.uniform = uni.data_ptr() if uni is not None else None
        nrm = t.randn(B, self.n_normal, generator=self._gen, device=self.device, dtype=t.float64) \
            if self.n_normal else None
        rio.normal = nrm.data_ptr() if nrm is not None else None
        if self.per_source:
            rio.mode = self.sampling_mode.data_ptr()                       # the sources the episode started with
        rio.x = self.x.data_ptr()
        rio.keep_state = 1
        rio.obs = self.buf['obs'].data_ptr()
        with t.cuda.device(self.device):
            capi.check(capi.lib().opfx_reset(self._env_handle, B, C.byref(rio), capi._stream()), 'opfx_reset')


class StochasticObservation:
    """wrappers/stochastic_obs.py:10-52 for the batch: uniform noise of
    `noise_relative_range` × (observation range) added to every observation, optionally
    clipped to the observation space."""

    def __init__(self, env, noise_relative_range: float = 0.1, maintain_original_range: bool = True):
        self.env = env
        self.maintain_original_range = maintain_original_range
        rng_ = env.observation_space.high - env.observation_space.low
        self.abs_noise_range = noise_relative_range * rng_
        self.observation_space = env.observation_space if maintain_original_range else \
            Box(env.observation_space.low - self.abs_noise_range, env.observation_space.high + self.abs_noise_range)

    def __getattr__(self, name):
        return getattr(self.env, name)

    def observation(self, obs):
        t = self.env.torch
        dev = obs.device
        rng_ = t.as_tensor(self.abs_noise_range, device=dev, dtype=obs.dtype)
        noise = (t.rand(obs.shape, generator=self.env._gen, device=dev, dtype=obs.dtype) * 2 - 1) * rng_
        obs = obs + noise
        if self.maintain_original_range:
            lo = t.as_tensor(self.observation_space.low, device=dev, dtype=obs.dtype)
            hi = t.as_tensor(self.observation_space.high, device=dev, dtype=obs.dtype)
            obs = t.minimum(t.maximum(obs, lo), hi)
        return obs

    def reset(self, **kw):
        obs, info = self.env.reset(**kw)
        return self.observation(obs), info

    def step(self, action):
        obs, reward, terminated, truncated, info = self.env.step(action)
        return self.observation(obs), reward, terminated, truncated, info
