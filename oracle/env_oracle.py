"""CPU oracle for the environment half of the hot path (SURVEY.md §8a rows
E2–E11): one instance at a time, plain numpy on the net's tables.

TEST INFRASTRUCTURE ONLY — imported by tests/, `__graft_entry__.smoke()` and
the `cpu_baseline` leg of bench.py, never by the product package.

Each function restates one piece of `/root/reference/opfgym/` (file:line cited
per function).  The oracle is PINNED: tests/test_oracle_env.py replays the
inputs stored in tests/golden/*.npz — produced by running the reference's own
environment classes (tests/golden/make_golden.py) — and requires the outputs to
match.  The power flow underneath is oracle/pf_oracle.py (parity unpinned
against pandapower, see there).
"""
from __future__ import annotations

import copy

import numpy as np

from . import pf_oracle


# ---------------------------------------------------------------------------
# E2: OpfEnv._set_simbench_state  (opf_env.py:317-372)
# ---------------------------------------------------------------------------
def set_simbench_state(net, profiles, step, noise=None, col_range=None, interp=None,
                       normal_noise_factor=0.0):
    """Gather profile row `step` for every profile table, optionally interpolate
    towards the next step (`interp`: one r per non-empty table, opf_env.py:345-349),
    apply noise — `noise` is the flat vector of multiplicative factors (:352-356),
    or of standard-normal draws z when `normal_noise_factor` > 0 (value +
    |value|·factor·z, :357-360) — clip to the column's min/max over the whole
    profile (:364-369) and write the unit table column (:371-372)."""
    off = 0
    k_tab = 0
    total = len(profiles[('load', 'q_mvar')])
    for (unit, col), df in profiles.items():
        n = df.shape[1]
        if not n:
            continue
        idx = net[unit].index
        data = df.loc[step, idx].to_numpy(float)
        if interp is not None and step < total - 1:
            r = interp[k_tab]
            data = data * r + df.loc[step + 1, idx].to_numpy(float) * (1 - r)
        k_tab += 1
        if noise is not None:
            if normal_noise_factor > 0:
                data = data + np.abs(data) * normal_noise_factor * noise[off:off + n]
            else:
                data = data * noise[off:off + n]
        off += n
        lo, hi = col_range[(unit, col)] if col_range else (df.min()[idx].to_numpy(float),
                                                          df.max()[idx].to_numpy(float))
        _assign(net, unit, col, idx, np.clip(data, lo, hi))


def profile_ranges(profiles):
    """Per-column min/max, computed once (the reference recomputes them on
    every reset, opf_env.py:364-369 — defect D10, numerically identical)."""
    return {k: (df.min().to_numpy(float), df.max().to_numpy(float))
            for k, df in profiles.items() if df.shape[1]}


def _assign(net, unit, col, idxs, values):
    """`net[unit].loc[idxs, col] = values` as the reference writes it (opf_env.py:284,315,368,483).  Where floats go
    into an integer column pandas upcasts the column today and will refuse tomorrow: the cast is made explicit."""
    df = net[unit]
    values = np.asarray(values)
    if values.dtype.kind == 'f' and df[col].dtype.kind in 'iub':
        df[col] = df[col].astype(float)
    df.loc[idxs, col] = values


# ---------------------------------------------------------------------------
# E3: `_sample_from_range` and the benchmark envs' `_sampling` tails
# ---------------------------------------------------------------------------
def sample_from_range(net, unit, col, idxs, draws):
    """opf_env.py:266-284: U[min_min|min, max_max|max] per row (`draws` are the
    U[0,1) numbers, consumed in order), divided by `scaling` where it exists."""
    df = net[unit]
    lo = (df[f'min_min_{col}'] if f'min_min_{col}' in df else df[f'min_{col}']).loc[idxs].to_numpy(float)
    hi = (df[f'max_max_{col}'] if f'max_max_{col}' in df else df[f'max_{col}']).loc[idxs].to_numpy(float)
    u = np.array([next(draws) for _ in range(len(idxs))])
    r = lo + (hi - lo) * u
    if 'scaling' in df:
        r = r / df['scaling'].loc[idxs].to_numpy(float)
    _assign(net, unit, col, idxs, r)


def sample_normal(net, state_keys, zdraws, relative_std=None):
    """opf_env.py:286-315 (`truncated=False`): N(mean, std·diff) clipped to the scaled
    min_min/max_max range; `zdraws` are standard-normal numbers consumed in order."""
    for unit, col, idxs in state_keys:
        if 'res_' in unit or 'poly_cost' in unit:
            continue
        df = net[unit].loc[idxs]
        hi = (df[f'max_max_{col}'] / df.scaling).to_numpy(float)
        lo = (df[f'min_min_{col}'] / df.scaling).to_numpy(float)
        diff = hi - lo
        std = relative_std * diff if relative_std else df[f'std_dev_{col}'].to_numpy(float)
        z = np.array([next(zdraws) for _ in range(len(idxs))])
        vals = df[f'mean_{col}'].to_numpy(float) + (std * diff) * z
        _assign(net, unit, col, idxs, np.clip(vals, lo, hi))


def sample_truncated_normal(net, state_keys, udraws, relative_std=None):
    """opf_env.py:304-307 (`truncated=True`): `stats.truncnorm.rvs(min_values, max_values, mean, std * diff)`.
    scipy's first two arguments are standardised bounds, so this is mean + std*diff * Z with Z a standard
    normal truncated to [min_values, max_values] (as written in the reference, D14).  scipy draws from its
    own generator; the oracle takes the uniform numbers `udraws` instead and applies the inverse CDF —
    `truncnorm.ppf(u, a, b, loc, scale)`, the transform `rvs` applies to its uniform draws."""
    from scipy import stats
    for unit, col, idxs in state_keys:
        if 'res_' in unit or 'poly_cost' in unit:
            continue
        df = net[unit].loc[idxs]
        hi = (df[f'max_max_{col}'] / df.scaling).to_numpy(float)
        lo = (df[f'min_min_{col}'] / df.scaling).to_numpy(float)
        diff = hi - lo
        std = relative_std * diff if relative_std else df[f'std_dev_{col}'].to_numpy(float)
        u = np.array([next(udraws) for _ in range(len(idxs))])
        _assign(net, unit, col, idxs, stats.truncnorm.ppf(u, lo, hi, df[f'mean_{col}'].to_numpy(float), std * diff))


def tail_voltage_control(net, draws, market_based):
    """voltage_control.py:111-133."""
    if market_based:
        for unit in ('sgen', 'ext_grid', 'storage'):
            idx = net.poly_cost[net.poly_cost.et == unit].index
            sample_from_range(net, 'poly_cost', 'cq2_eur_per_mvar2', idx, draws)
    for unit in ('sgen', 'storage'):
        df = net[unit]
        df['max_p_mw'] = df.p_mw * df.scaling + 1e-9
        df['min_p_mw'] = df.p_mw * df.scaling - 1e-9
    for unit in ('sgen', 'storage'):
        df = net[unit]
        q_max = (df.max_s_mva ** 2 - df.max_p_mw ** 2) ** 0.5
        df['min_q_mvar'] = -q_max
        df['max_q_mvar'] = q_max
        df['q_mvar'] = 0.0


def tail_mixed_continuous_discrete(net, draws):
    """examples/mixed_continuous_discrete.py:99-110."""
    sample_from_range(net, 'ext_grid', 'vm_pu', net.ext_grid.index, draws)
    df = net['sgen']
    df['max_p_mw'] = df.p_mw * df.scaling + 1e-9
    df['min_p_mw'] = df.p_mw * df.scaling - 1e-9


def tail_custom_constraint(net, draws):
    """examples/custom_constraint.py:74-80."""
    df = net['sgen']
    df['max_p_mw'] = df.p_mw * df.scaling + 1e-9
    df['min_p_mw'] = df.p_mw * df.scaling - 1e-9


def tail_eco_dispatch(net, draws):
    """eco_dispatch.py:111-123."""
    sample_from_range(net, 'poly_cost', 'cp1_eur_per_mw', net.poly_cost.index, draws)
    sample_from_range(net, 'pwl_cost', 'cp1_eur_per_mw', net.pwl_cost.index, draws)
    for idx in net.ext_grid.index:
        net.pwl_cost.at[idx, 'points'] = [[0, 10000, net.pwl_cost.at[idx, 'cp1_eur_per_mw']]]


def tail_max_renewable(net, draws):
    """max_renewable.py:101-105."""
    net.sgen['max_p_mw'] = net.sgen.p_mw * net.sgen.scaling + 1e-6


def tail_load_shedding(net, draws, efficiency=0.95):
    """load_shedding.py:122-149."""
    sample_from_range(net, 'poly_cost', 'cp1_eur_per_mw', net.poly_cost.index, draws)
    sample_from_range(net, 'pwl_cost', 'cp1_eur_per_mw', net.pwl_cost.index, draws)
    for idx in net.pwl_cost.index:
        price = net.pwl_cost.at[idx, 'cp1_eur_per_mw']
        net.pwl_cost.at[idx, 'points'] = [[-1000, 0, price * efficiency], [0, 1000, price / efficiency]]
    net.load['max_p_mw'] = net.load['p_mw'] * net.load.scaling + 1e-9
    for unit in ('load', 'storage'):
        net[unit]['max_q_mvar'] = net[unit].q_mvar * net[unit].scaling + 1e-9
        net[unit]['min_q_mvar'] = net[unit].q_mvar * net[unit].scaling - 1e-9


TAILS = {'VoltageControl': lambda net, d: tail_voltage_control(net, d, False),
         'QMarket': lambda net, d: tail_voltage_control(net, d, True),
         'EcoDispatch': tail_eco_dispatch, 'EcoDispatchSharedBus': tail_eco_dispatch, 'MaxRenewable': tail_max_renewable,
         'SecurityConstrained': lambda net, d: None, 'LoadShedding': tail_load_shedding,
         'MultiStageOpf': lambda net, d: None, 'NetworkReconfiguration': lambda net, d: None,
         'SwitchedShunts': lambda net, d: None, 'BusbarCouplers': lambda net, d: None,
         'MixedContinuousDiscrete': tail_mixed_continuous_discrete,
         'ConstraintSatisfaction': lambda net, d: None, 'PartiallyObservable': lambda net, d: None,
         'NonSimbenchNet': lambda net, d: None, 'AddCustomConstraint': tail_custom_constraint}


# ---------------------------------------------------------------------------
# E4: OpfEnv._apply_actions / get_current_actions  (opf_env.py:421-491, 566-588)
# ---------------------------------------------------------------------------
def apply_actions(net, act_keys, action, autoscale=True, diff_step=None):
    action = np.clip(np.asarray(action, float), 0.0, 1.0)                # :429
    k = 0
    for unit, col, idxs in act_keys:
        n = len(idxs)
        if n == 0:
            continue
        df = net[unit]
        a = action[k:k + n]
        pre = ('min_', 'max_') if autoscale else ('min_min_', 'max_max_')   # :439-446
        lo = df[pre[0] + col].loc[idxs].to_numpy(float)
        hi = df[pre[1] + col].loc[idxs].to_numpy(float)
        sc = df['scaling'].loc[idxs].to_numpy(float) if 'scaling' in df.columns else np.ones(n)
        if diff_step:                                                       # :451-458
            prev = df[col].loc[idxs].to_numpy(float) * sc
            sp = (a * 2 - 1) * diff_step * (hi - lo) + prev
        else:
            sp = a * (hi - lo) + lo                                         # :461
        if not autoscale or diff_step:                                      # :464-470
            # (masked assignment as in the reference: a NaN bound — e.g. sqrt of a negative reactive
            # headroom — compares False and leaves the set-point alone; np.minimum would propagate it)
            with np.errstate(invalid='ignore'):
                if f'max_{col}' in df.columns:
                    mx = df[f'max_{col}'].loc[idxs].to_numpy(float)
                    sp = np.where(sp > mx, mx, sp)
                if f'min_{col}' in df.columns:
                    mn = df[f'min_{col}'].loc[idxs].to_numpy(float)
                    sp = np.where(sp < mn, mn, sp)
        sp = sp / sc                                                        # :472-474
        if col in ('closed', 'in_service'):                                 # :476-478
            sp = np.round(sp).astype(bool)
        elif col in ('tap_pos', 'step'):                                    # :479-481
            sp = np.round(sp)
        _assign(net, unit, col, idxs, sp)                                   # :483
        k += n
    cur = current_actions(net, act_keys, autoscale, from_results=False)
    with np.errstate(invalid='ignore'):
        return float(np.mean(np.abs(cur - action))) if len(action) else 0.0   # :488-489


def current_actions(net, act_keys, autoscale=True, from_results=True):
    out = []
    for unit, col, idxs in act_keys:
        df = net[unit]
        if from_results:
            sp = net['res_' + unit][col].loc[idxs].to_numpy(float)
        else:
            sp = df[col].loc[idxs].to_numpy(float)
            if 'scaling' in df.columns:
                sp = sp * df['scaling'].loc[idxs].to_numpy(float)
        pre = ('min_', 'max_') if autoscale else ('min_min_', 'max_max_')
        lo = df[pre[0] + col].loc[idxs].to_numpy(float)
        hi = df[pre[1] + col].loc[idxs].to_numpy(float)
        with np.errstate(invalid='ignore', divide='ignore'):
            out.append((sp - lo) / (hi - lo))
    return np.concatenate(out) if out else np.zeros(0)


# ---------------------------------------------------------------------------
# E6: objective.get_pandapower_costs  (objective.py:6-87)
# ---------------------------------------------------------------------------
def cost_vector(net):
    """[poly P costs…, poly Q costs…, pwl costs…] (objective.py:28,45)."""
    parts = []
    pc = net.poly_cost
    if len(pc):
        p = np.array([net['res_' + et][col].loc[el] for et, el, col in
                      zip(pc.et, pc.element, ['p_mw'] * len(pc))], float)
        q = np.array([net['res_' + et]['q_mvar'].loc[el] for et, el in zip(pc.et, pc.element)], float)
        pcost = pc.cp0_eur.to_numpy(float) + pc.cp1_eur_per_mw.to_numpy(float) * p \
            + pc.cp2_eur_per_mw2.to_numpy(float) * p ** 2
        qcost = pc.cq0_eur.to_numpy(float) + pc.cq1_eur_per_mvar.to_numpy(float) * q \
            + pc.cq2_eur_per_mvar2.to_numpy(float) * q ** 2
        parts += [pcost, qcost]
    pw = net.pwl_cost
    if len(pw):
        power = np.array([net['res_' + et]['p_mw' if pt == 'p' else 'q_mvar'].loc[el]
                          for et, el, pt in zip(pw.et, pw.element, pw.power_type)], float)
        costs = np.zeros(len(pw))
        nseg = min(len(p) for p in pw.points)                # zip(*points) truncation, defect D9
        for s in range(nseg):                                               # :60-75
            lo = np.array([p[s][0] for p in pw.points], float)
            hi = np.array([p[s][1] for p in pw.points], float)
            price = np.array([p[s][2] for p in pw.points], float)
            sign = np.sign(power)
            same = sign == np.sign(lo + hi)
            inside = np.minimum(np.abs(lo), np.abs(hi))
            in_flag = (np.abs(power) > inside) & same
            out_flag = np.abs(power) > np.maximum(np.abs(lo), np.abs(hi))
            mid = in_flag & ~out_flag
            costs[out_flag] += (sign * (hi - lo) * price)[out_flag]
            costs[mid] += (sign * (np.abs(power) - inside) * price)[mid]
        parts.append(costs)
    return np.concatenate(parts) if parts else np.zeros(0)


# ---------------------------------------------------------------------------
# E7: Constraint.get_violation_metrics  (constraints.py:70-128)
# ---------------------------------------------------------------------------
def violation_metrics(net, con):
    """`con`: any object with the reference Constraint's attributes
    (unit_type, values_column, only_worst_case_violations, autoscale_violation,
    scale_bounded_values, penalty_factor, penalty_power, violation_count_penalty)."""
    get_values = getattr(con, 'get_values', None)                           # constraints.py:62-63
    values = np.asarray(get_values(net), float) if get_values is not None else \
        net['res_' + con.unit_type][con.values_column].to_numpy(float)
    tbl = net[con.unit_type]
    violation, n_viol = 0.0, 0
    custom_bounds = getattr(con, 'get_boundaries_fn', None)                 # constraints.py:64-65
    custom_bounds = custom_bounds(net) if custom_bounds is not None else None
    for which in ('min', 'max'):                                            # :93-98
        col = f'{which}_{con.values_column}'
        if custom_bounds is not None:
            if which not in custom_bounds:
                continue
            bound = np.asarray(custom_bounds[which], float)
        else:
            if col not in tbl:
                continue
            bound = tbl[col].to_numpy(float)
            if con.scale_bounded_values or ('scaling' in tbl and con.values_column in ('p_mw', 'q_mvar')):
                bound = bound * tbl['scaling'].to_numpy(float)              # :104-108
        invalid = values > bound if which == 'max' else values < bound      # :110-111
        n_viol += int(invalid.sum())
        if invalid.any():                                                   # :113-122
            absv = np.abs(values - bound)[invalid]
            violation += absv.max() if con.only_worst_case_violations else absv.sum()
    autoscale = con.autoscale_violation
    if not autoscale and con.unit_type == 'ext_grid':                       # :179-182, 189-192
        autoscale = 1 / abs(net.ext_grid['mean_' + con.values_column].sum())
    if autoscale:
        violation = violation * autoscale                                   # :82-83 (True -> ×1, defect D8)
    penalty = -(violation ** con.penalty_power * con.penalty_factor
                + n_viol * con.violation_count_penalty)                     # :124-128
    return n_viol == 0, violation, penalty


# ---------------------------------------------------------------------------
# E8: reward  (reward.py:61-98, 219-320)
# ---------------------------------------------------------------------------
def reward_and_cost(rf, objective, penalty, valid):
    """`rf`: dict(kind, penalty_weight, clip_range, scaling_params, valid_reward,
    invalid_penalty, invalid_objective_share)."""
    kind = rf.get('kind', 'summation')
    obj, pen = objective, penalty
    if kind == 'replacement':
        obj = obj + rf['valid_reward'] if valid else 0.0                    # :247-252
    elif kind == 'parameterized':
        pen = pen + rf['valid_reward'] if valid else pen - rf['invalid_penalty']   # :288-291
        if not valid:
            obj = obj * rf['invalid_objective_share']                       # :293-298
    elif kind == 'onlyobjective':
        pen = 0.0                                                           # :316-317
    sp = rf.get('scaling_params') or {'objective_factor': 1, 'objective_bias': 0,
                                      'penalty_factor': 1, 'penalty_bias': 0}
    obj = obj * sp['objective_factor'] + sp['objective_bias']               # :83-91
    pen = pen * sp['penalty_factor'] + sp['penalty_bias']
    w = rf.get('penalty_weight', 0.5)
    reward = obj + pen if w is None else obj * (1 - w) + pen * w            # :78-81
    if rf.get('clip_range'):
        reward = float(np.clip(reward, *rf['clip_range']))                  # :70-71
    cost = 0.0 if valid else abs(penalty * sp['penalty_factor'])            # :93-98
    if kind == 'parameterized' and not valid:
        cost += rf['invalid_penalty']                                       # :301-305
    return reward, cost


# ---------------------------------------------------------------------------
# E9: OpfEnv._get_obs  (opf_env.py:532-549)
# ---------------------------------------------------------------------------
def observation(net, obs_keys, add_mean_obs=False, time_obs=None, bus_wise_obs=False):
    parts = []
    for unit, col, idxs in obs_keys:
        if unit == 'load' and bus_wise_obs:                                 # :535-536, 806-810
            df = net[unit].iloc[np.asarray(idxs)]
            parts.append(df.groupby(['bus'])[col].sum().to_numpy(float))
        else:
            parts.append(net[unit].loc[idxs, col].to_numpy(float))
    if add_mean_obs:
        parts.append(np.array([np.mean(p) for p in parts if len(p) > 1]))   # :539-542
    if time_obs is not None:
        parts = [np.asarray(time_obs, float)] + parts                       # :544-547
    return np.concatenate(parts) if parts else np.zeros(0)


# ---------------------------------------------------------------------------
# E10/E11: step and the N-1 loop
# ---------------------------------------------------------------------------
class EnvOracle:
    """One instance; mirrors reset()/step() of opf_env.py:177-220, 374-419 and
    SecurityConstrainedOpfEnv.calculate_violations (security_constrained.py:37-68)."""

    def __init__(self, net, act_keys, obs_keys, profiles, constraints, reward, tail, *,
                 autoscale_actions=True, diff_action_step_size=None, clipped_action_penalty=0.0,
                 diff_objective=False, add_mean_obs=False, pf_for_obs=False, steps_per_episode=1,
                 n_minus_one_keys=(), not_converged_penalty=1, enforce_q_lims=True,
                 data='simbench', state_keys=None, sampling_params=None, bus_wise_obs=False,
                 multi_stage=False, split=None, objective=None):
        self.objective_fn = objective            # `objective_function(net)` seam (opf_env.py:80-84)
        self.base_net = net
        # carry_over=True keeps the net of the previous episode at reset, as the reference's single env does:
        # a column that one data source samples and another does not (e.g. gen.p_mw set by the profile row
        # but not by _sample_uniform when it is no state key) then leaks into the next episode (defect D12).
        # Default False: every reset starts from the base net (what the batched product does).
        self.carry_over = False
        self.net = copy.deepcopy(net)
        self.act_keys, self.obs_keys = act_keys, obs_keys
        self.profiles, self.ranges = profiles, profile_ranges(profiles) if profiles else None
        self.constraints, self.reward, self.tail = constraints, reward, tail
        self.autoscale, self.diff_step = autoscale_actions, diff_action_step_size
        self.cap = clipped_action_penalty
        self.diff_objective, self.add_mean_obs, self.pf_for_obs = diff_objective, add_mean_obs, pf_for_obs
        self.spe = steps_per_episode
        self.n1, self.ncp = n_minus_one_keys, not_converged_penalty
        self.enforce_q_lims = enforce_q_lims
        self.data, self.state_keys = data, state_keys
        self.sampling_params = sampling_params or {}
        self.bus_wise_obs, self.multi_stage, self.split = bus_wise_obs, multi_stage, split
        self.initial_obj = 0.0
        # start of every power flow: 'flat' (what the product runs by default) or 'dc' — what pandapower's default
        # init='auto' resolves to on grids fed above 70 kV; same solution, other iteration counts
        self.init = 'flat'

    def costs(self):
        return np.asarray(self.objective_fn(self.net), float) if self.objective_fn else cost_vector(self.net)

    def solve(self):
        """One `run_power_flow` (opf_env.py:646-662).  `solve_iterations` collects the Newton iterations of every
        successful call since the last reset()/step() began (base case first, then the contingencies)."""
        try:
            sol = pf_oracle.runpp(self.net, enforce_q_lims=self.enforce_q_lims, init=self.init)
            self.solve_iterations.append(int(sol['iterations']))
            return True
        except pf_oracle.LoadflowNotConverged:
            return False

    def reset(self, step, uniform=(), noise=None, initial_action=None, interp=None, normal=(), data=None):
        """`data`: the distribution to sample from when it is not the training one (the reference picks
        test_data for reset(options={'test': True}), opf_env.py:226)."""
        if not (self.carry_over and getattr(self, 'net', None) is not None):
            self.net = copy.deepcopy(self.base_net)
        self.solve_iterations = []
        self.step_in_episode = 0
        self.current_step = step
        draws = iter(np.asarray(uniform, float))
        data = data or self.data
        if 'noise_factor' in self.sampling_params and data != 'mixed':      # opf_env.py:231 comes first
            data = 'noisy_simbench'
        if data == 'mixed' and 'noise_factor' not in self.sampling_params:  # opf_env.py:242-251
            r = float(np.asarray(interp, float).ravel()[0])                 # the first draw of the reset (:244)
            probs = self.sampling_params.get('data_probabilities', (0.5, 0.75, 1.0))
            data = 'noisy_simbench' if r < probs[0] else ('full_uniform' if r < probs[1] else 'normal_around_mean')
            interp = None
        if data == 'full_uniform':                                          # opf_env.py:238-239, 253-264
            for unit, col, idxs in self.state_keys:
                if 'res_' not in unit:
                    sample_from_range(self.net, unit, col, idxs, draws)
        elif data == 'normal_around_mean' and self.sampling_params.get('truncated'):   # :240-241, 304-307
            sample_truncated_normal(self.net, self.state_keys, draws, self.sampling_params.get('relative_std'))
        elif data == 'normal_around_mean':                                  # :240-241
            sample_normal(self.net, self.state_keys, iter(np.asarray(normal, float)),
                          self.sampling_params.get('relative_std'))
        else:
            nnf = self.sampling_params.get('noise_factor', 0.1) \
                if self.sampling_params.get('noise_distribution') == 'normal' else 0.0
            set_simbench_state(self.net, self.profiles, step, noise, self.ranges, interp, nnf)
        self.tail(self.net, draws)
        n_act = sum(len(i) for _, _, i in self.act_keys)
        act = np.full(n_act, 0.5) if initial_action is None else initial_action   # opf_env.py:201-207
        apply_actions(self.net, self.act_keys, act, self.autoscale, None)
        if self.pf_for_obs:                                                 # :209-216
            assert self.solve()
            self.initial_obj = float(np.sum(-self.costs()))
        return observation(self.net, self.obs_keys, self.add_mean_obs, bus_wise_obs=self.bus_wise_obs)

    def violations(self):
        res = [violation_metrics(self.net, c) for c in self.constraints]
        return (np.array([r[0] for r in res], bool), np.array([r[1] for r in res], float),
                np.array([r[2] for r in res], float))

    def step(self, action):
        self.step_in_episode += 1
        self.solve_iterations = []
        corr = apply_actions(self.net, self.act_keys, action, self.autoscale, self.diff_step)
        if not self.solve():
            return dict(converged=False)
        objective = float(np.sum(-self.costs()))                            # :493-500, 517
        if self.diff_objective:
            objective -= self.initial_obj
        valids, viol, pen = self.violations()
        for unit, col, idxs in self.n1:                                     # security_constrained.py:44-66
            for idx in idxs:
                if not bool(self.net[unit].at[idx, col]):
                    continue
                self.net[unit].at[idx, col] = False
                if self.solve():
                    v2, vi2, p2 = self.violations()
                    valids, viol, pen = valids & v2, viol + vi2, pen + p2
                else:
                    valids = np.zeros_like(valids)
                    viol = viol + self.ncp
                    pen = pen + self.ncp                                    # sign as in the reference (D6)
                self.net[unit].at[idx, col] = True
        valid = bool(valids.all())
        reward, cost = reward_and_cost(self.reward, objective, float(np.sum(pen)), valid)
        if self.cap:
            reward -= corr * self.cap                                       # opf_env.py:403-404
        term = self.spe == 1
        trunc = (not term) and self.step_in_episode >= self.spe
        obs = observation(self.net, self.obs_keys, self.add_mean_obs, bus_wise_obs=self.bus_wise_obs)
        if self.multi_stage:                                                # multi_stage.py:26-58
            test_steps, val_steps, _ = self.split
            new_step = self.current_step + 1
            if new_step in set(val_steps.tolist()) or new_step in set(test_steps.tolist()):
                trunc = True                                                # training mode (:36-39)
            if self.step_in_episode >= self.spe:
                term = True                                                 # :42-43
            if not (term or trunc):
                self.current_step = new_step
                set_simbench_state(self.net, self.profiles, new_step, None, self.ranges)   # :50
                self.tail(self.net, iter(()))
                if self.pf_for_obs:                                         # :52-53
                    self.solve()
                obs = observation(self.net, self.obs_keys, False)           # :56 (no mean obs there)
        return dict(converged=True, obs=obs,
                    reward=reward, terminated=term, truncated=trunc, valids=valids, violations=viol,
                    penalties=pen, cost=cost, objective=objective, mean_correction=corr,
                    vm_pu=self.net.res_bus.vm_pu.to_numpy(float),
                    va_degree=self.net.res_bus.va_degree.to_numpy(float),
                    line_loading=self.net.res_line.loading_percent.to_numpy(float),
                    trafo_loading=self.net.res_trafo.loading_percent.to_numpy(float),
                    p_ext=self.net.res_ext_grid.p_mw.to_numpy(float),
                    q_ext=self.net.res_ext_grid.q_mvar.to_numpy(float),
                    **({'q_gen': self.net.res_gen.q_mvar.to_numpy(float)} if len(self.net.gen) else {}),
                    **({'trafo3w_loading': self.net.res_trafo3w.loading_percent.to_numpy(float)}
                       if 'trafo3w' in self.net and len(self.net.trafo3w) and 'res_trafo3w' in self.net else {}))


# ---------------------------------------------------------------------------
# E12: estimate_reward_distribution  (reward.py:181-216)
# ---------------------------------------------------------------------------
def estimate_reward_distribution(orc: EnvOracle, steps, uniform, noise, actions) -> dict:
    """The twelve statistics of reward.py:198-216 over explicit samples: per sample a reset at `steps[k]`
    with the given draws (:186), the action applied as absolute set-points (:188, `_apply_actions` without a
    step size), one power flow (:189), Σ(-costs) (:190) and Σ penalties incl. the N-1 loop (:191); rows
    whose power flow failed are dropped (:197-198)."""
    objectives, penalties = [], []
    saved = orc.diff_step, orc.diff_objective
    orc.diff_step, orc.diff_objective = None, False
    try:
        for k in range(len(steps)):
            orc.reset(int(steps[k]), uniform[k] if uniform is not None else (), noise[k] if noise is not None else None)
            out = orc.step(actions[k])
            objectives.append(out['objective'] if out['converged'] else np.nan)
            penalties.append(float(np.sum(out['penalties'])) if out['converged'] else np.nan)
    finally:
        orc.diff_step, orc.diff_objective = saved
    o, p = np.array(objectives), np.array(penalties)
    o, p = o[~np.isnan(o)], p[~np.isnan(p)]
    return {'min_objective': o.min(), 'max_objective': o.max(), 'min_penalty': p.min(), 'max_penalty': p.max(),
            'mean_objective': o.mean(), 'mean_penalty': p.mean(), 'std_objective': np.std(o), 'std_penalty': np.std(p),
            'median_objective': np.median(o), 'median_penalty': np.median(p),
            'mean_abs_objective': np.abs(o).mean(), 'mean_abs_penalty': np.abs(p).mean()}
