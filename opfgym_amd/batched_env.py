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
from .case import REF, expand_dclines, static_consumption
from .simbench_build import define_test_train_split, get_simbench_time_observation
from .descriptors import DescriptorCompiler, _case_all_branches_in
from .spaces import Box, get_obs_and_state_space  # noqa: F401
from .store import ColumnStore, OpsBuilder  # noqa: F401  (envs.py and the tests import them from here)
from .topology import TopologyMixin

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


class BatchedOpfEnv(DescriptorCompiler, TopologyMixin):
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
        # (DC lines run as two generators each, as in pandapower: from here on they ARE rows at the end of the generator table —
        #  static ones, nothing samples or actuates them — and the dcline table of the environment's copy of the net is empty)
        if expand_dclines(net) is not net:
            net = expand_dclines(net)
            net['dcline'] = net['dcline'].iloc[0:0]
            del net['_dcline_gens']
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
        static = {(tbl, col): v for tbl, pq in static_consumption(net).items() for col, v in zip(('_p_mw', '_q_mvar'), pq)}
        for (tbl, col), v in static.items():                    # (wards, motors: computed columns, descriptors.py)
            self.store.computed(tbl, col, v)
        # sampling programme first: it decides which columns are per-instance
        self._build_sampling()
        # ... and once more with the columns the STEP never reads laid out last (intermediates of the reset programme such
        # as sgen.max_p_mw of voltage_control.py:123-125, limit columns of nothing that acts): opfx_step stages the row up
        # to the last column a descriptor names (opfx_env_get_row_io), what lies behind stays in HBM
        order = sorted(self.store.ranges, key=lambda k: (k not in self._step_columns(), self.store.ranges[k][0]))
        if order != sorted(self.store.ranges, key=lambda k: self.store.ranges[k][0]):
            self.store = ColumnStore(net)
            for tbl, col in order:
                if (tbl, col) in static:
                    self.store.computed(tbl, col, static[(tbl, col)])
                else:
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

    def _alloc(self, B):
        t, dev = self.torch, self.device
        f64 = dict(dtype=t.float64, device=dev)
        u8 = dict(dtype=t.bool, device=dev)          # one byte each; the kernel writes 0/1
        nc = max(1, self.n_device_constraints)      # (host constraints get their columns in host_fallback.finish)
        self.B = B
        self._state_valid = False
        # (the library's scratch rows for launches of this batch: allocated here, not by the first step — ADVICE r05)
        capi.check(capi.lib().opfx_env_prepare(self._env_handle, int(B)), 'opfx_env_prepare')
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
