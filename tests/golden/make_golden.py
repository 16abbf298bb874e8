"""Generate tests/golden/*.npz from the REFERENCE's Python layer.

Runs only in the build container (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/make_golden.py

`opfgym` (the reference) is imported from /root/reference with the throw-away
stub packages of tests/golden/_stubs standing in for gymnasium / pandapower /
simbench: the grid comes from this repo's synthetic generator and every
`pp.runpp` call is answered by the SciPy oracle (oracle/pf_oracle.py).  So the
fixtures pin the reference's OWN environment logic — `_define_opf`, `_sampling`,
`_set_simbench_state`, `_apply_actions`, objective, constraints, reward, info,
observation, N-1 loop — on top of the oracle's power flow.  Only inputs/outputs
are stored (no reference source text).
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.environ.get("OPFX_GOLDEN_OUT") or HERE          # (tests regenerate into a scratch directory)
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(HERE, '_stubs'), ROOT, '/root/reference', HERE]

import numpy as np  # noqa: E402

import opfgym.envs  # noqa: E402,F401  (reference)
import opfgym.examples.security_constrained as ref_sc_example  # noqa: E402
import opfgym.security_constrained as ref_sc  # noqa: E402
from scenarios import (CANDIDATE_STEPS, E12_SCENARIOS, EPISODE_START_STEPS, EPISODE_STEPS, FAILED_CONTINGENCY_ROWS,  # noqa: E402
                       SCENARIOS, TRACKED, VALID_ROWS)
import opfgym.examples.multi_stage as ref_ms  # noqa: E402
import opfgym.examples.network_reconfiguration as ref_nr  # noqa: E402
import opfgym.examples.mixed_continuous_discrete as ref_mcd  # noqa: E402
import opfgym.examples.pure_constraint_satisfaction as ref_cs  # noqa: E402
import opfgym.examples.partial_obs as ref_po  # noqa: E402
import opfgym.examples.non_simbench_net as ref_ns  # noqa: E402
import opfgym.examples.custom_constraint as ref_cc  # noqa: E402
import opfgym.constraints as ref_constraints  # noqa: E402
import opfgym.opf_env as ref_opf_env  # noqa: E402


class RefAddCustomConstraint(ref_opf_env.OpfEnv):
    """examples/custom_constraint.py with its constraint list handed to the parameter OpfEnv
    actually reads (`custom_constraints`; the example's `constraints=` keyword is swallowed): the
    reference's own Constraint class with the example's value/boundary callables."""

    def __init__(self, simbench_network_name, cos_phi=0.95, **kwargs):
        self.cos_phi = cos_phi
        net, profiles = ref_cc.AddCustomConstraint._define_opf(self, simbench_network_name)
        obs_keys = [('load', 'p_mw', net.load.index), ('load', 'q_mvar', net.load.index)]
        act_keys = [('sgen', 'q_mvar', net.sgen.index)]
        cl = ref_constraints.create_default_constraints(net, {})
        cl.append(ref_constraints.Constraint('sgen', 's_mva', get_values=ref_cc.get_s_mva_values,
                                             get_boundaries=ref_cc.get_s_mva_boundaries))
        super().__init__(net, act_keys, obs_keys, profiles=profiles, optimal_power_flow_solver=False,
                         custom_constraints=cl, **kwargs)

    def _sampling(self, *args, **kwargs):            # examples/custom_constraint.py:74-80
        super()._sampling(*args, **kwargs)
        self.net.sgen['max_p_mw'] = self.net.sgen.p_mw * self.net.sgen.scaling + 1e-9
        self.net.sgen['min_p_mw'] = self.net.sgen.p_mw * self.net.sgen.scaling - 1e-9


class RefScVoltageControl(opfgym.envs.VoltageControl, ref_sc.SecurityConstrainedOpfEnv):
    """BASELINE config 5 on the reference side: the reference's own VoltageControl (its `_define_opf`,
    `_sampling`, keys) composed with the reference's own SecurityConstrainedOpfEnv (its N-1
    `calculate_violations`) by inheritance — no reference code is restated here.  The stand-in grid's PV
    generators become fixed sgens first (VoltageControl asserts a grid without `gen` rows), with the same
    helper the product uses; contingencies: every in-service line whose outage does not island."""

    def __init__(self, simbench_network_name, n_minus_one_lines='all', **kwargs):
        import opfgym.envs.voltage_control as vc_mod
        from opfgym_amd.simbench_build import gens_to_fixed_sgens, non_islanding_lines
        original = vc_mod.build_simbench_net
        chosen = {}

        def prepared(*a, **k):
            net, profiles = original(*a, **k)
            gens_to_fixed_sgens(net, profiles)
            chosen['lines'] = non_islanding_lines(net) if isinstance(n_minus_one_lines, str) \
                else np.array(n_minus_one_lines)
            return net, profiles

        class _Keys(tuple):          # resolved once the net exists (VoltageControl builds it inside __init__)
            def __iter__(self_inner):
                return iter((('line', 'in_service', chosen['lines']),))
        vc_mod.build_simbench_net = prepared
        try:
            super().__init__(simbench_network_name, n_minus_one_keys=_Keys(), **kwargs)
        finally:
            vc_mod.build_simbench_net = original


class RefBusbarCouplers(ref_nr.NetworkReconfiguration):
    """The reference's own NetworkReconfiguration, unchanged, on a grid with two busbar couplers (the product's stand-in
    grid helper `simbench_build.split_busbars`); which switches are controllable is its own constructor argument."""

    def __init__(self, *args, **kwargs):
        from opfgym_amd.simbench_build import split_busbars
        original = ref_nr.build_simbench_net

        def prepared(*a, **k):
            net, profiles = original(*a, **k)
            split_busbars(net, profiles)
            return net, profiles
        ref_nr.build_simbench_net = prepared
        try:
            super().__init__(*args, **kwargs)
        finally:
            ref_nr.build_simbench_net = original


class RefSwitchedShunts(ref_nr.NetworkReconfiguration):
    """The reference's own NetworkReconfiguration on a grid with three shunts in steps, whose `('shunt', 'step')` key joins
    its action keys: the grid helper is the product's (`simbench_build.add_switched_shunts`, a stand-in grid matter), the
    key is appended to what the reference class hands to `OpfEnv.__init__` — no reference code is restated."""

    def __init__(self, *args, **kwargs):
        from opfgym_amd.simbench_build import add_switched_shunts
        original, base_init = ref_nr.build_simbench_net, ref_opf_env.OpfEnv.__init__

        def prepared(*a, **k):
            net, profiles = original(*a, **k)
            add_switched_shunts(net, profiles)
            return net, profiles

        def init_with_shunts(self_, net, act_keys, obs_keys, *a, **k):
            return base_init(self_, net, list(act_keys) + [('shunt', 'step', net.shunt.index)], obs_keys, *a, **k)
        ref_nr.build_simbench_net, ref_opf_env.OpfEnv.__init__ = prepared, init_with_shunts
        try:
            super().__init__(*args, **kwargs)
        finally:
            ref_nr.build_simbench_net, ref_opf_env.OpfEnv.__init__ = original, base_init


class RefEcoDispatchSharedBus(opfgym.envs.EcoDispatch):
    """The reference's own EcoDispatch on a grid whose generators share buses: the grid helper and the reactive ranges /
    prices are the product's stand-in helpers (`simbench_build.share_generator_buses`, `shared_bus_reactive_setup` — grid
    data, applied before / after the reference's `_define_opf`); objective, constraints and `res_gen` handling are the
    reference's, the per-generator reactive power comes from the oracle's pfsoln."""

    def _define_opf(self, simbench_network_name, *args, **kwargs):
        import opfgym.envs.eco_dispatch as ed_mod
        from opfgym_amd.simbench_build import share_generator_buses, shared_bus_reactive_setup
        original = ed_mod.build_simbench_net

        def prepared(*a, **k):
            net, profiles = original(*a, **k)
            share_generator_buses(net, profiles)
            return net, profiles
        ed_mod.build_simbench_net = prepared
        try:
            net, profiles = super()._define_opf(simbench_network_name, *args, **kwargs)
        finally:
            ed_mod.build_simbench_net = original
        shared_bus_reactive_setup(net)
        return net, profiles


REF = {'VoltageControl': opfgym.envs.VoltageControl, 'EcoDispatchSharedBus': RefEcoDispatchSharedBus, 'SecurityConstrainedVoltageControl': RefScVoltageControl, 'QMarket': opfgym.envs.QMarket,
       'EcoDispatch': opfgym.envs.EcoDispatch, 'MaxRenewable': opfgym.envs.MaxRenewable,
       'SecurityConstrained': ref_sc_example.SecurityConstrained, 'LoadShedding': opfgym.envs.LoadShedding, 'MultiStageOpf': ref_ms.MultiStageOpf,
       'NetworkReconfiguration': ref_nr.NetworkReconfiguration, 'SwitchedShunts': RefSwitchedShunts,
       'BusbarCouplers': RefBusbarCouplers,
       'MixedContinuousDiscrete': ref_mcd.MixedContinuousDiscrete,
       'ConstraintSatisfaction': ref_cs.ConstraintSatisfaction, 'PartiallyObservable': ref_po.PartiallyObservable,
       'NonSimbenchNet': ref_ns.NonSimbenchNet, 'AddCustomConstraint': RefAddCustomConstraint}


def snapshot(net):
    out = {}
    for tbl, col in TRACKED:
        if tbl in net and col in net[tbl].columns and len(net[tbl]):
            out[f'tab__{tbl}__{col}'] = np.array(net[tbl][col].to_numpy(dtype=float), copy=True)
    return out


def run_episodes(name):
    """Scenarios with several steps per episode: arrays get a step axis [n, S, ...]."""
    cls, kwargs, n, seed = SCENARIOS[name]
    S = EPISODE_STEPS[name]
    env = REF[cls](seed=seed, **kwargs)
    rng = np.random.default_rng(1000 + seed)
    rec = {}

    def push(key, val):
        rec.setdefault(key, []).append(np.array(val, copy=True))
    starts = EPISODE_START_STEPS.get(name)
    for k in range(n):
        step = int(starts[k]) if starts else int(rng.choice(env.train_steps))
        obs0, _ = env.reset(seed=seed * 100 + k, options={'step': step})
        log = env.np_random.log
        uni = [u.ravel() for kind, u in log if kind == 'uniform']
        push('step', step)
        push('uniform', np.concatenate(uni) if uni else np.zeros(0))
        push('noise', np.zeros(0))
        push('obs_reset', obs0)
        ep = {}
        done_at = S
        for s_ in range(S):
            action = rng.random(env.action_space.shape[0])
            if s_ >= done_at:                      # episode over: pad with NaN rows of the right shape
                for key in ep:
                    ep[key].append(np.full_like(np.asarray(ep[key][0], dtype=float), np.nan))
                continue
            obs, reward, terminated, truncated, info = env.step(action)
            assert 'cost' in info, 'multi-step golden episodes must converge'
            for key, val in (('action', action), ('obs_step', obs), ('reward', reward),
                             ('terminated', terminated), ('truncated', truncated), ('valids', info['valids']),
                             ('violations', info['violations']), ('penalties', info['unscaled_penalties']),
                             ('cost', info['cost']), ('vm_pu', env.net.res_bus.vm_pu.to_numpy()),
                             ('current_actions', env.get_current_actions())):
                ep.setdefault(key, []).append(np.array(val, dtype=float, copy=True))
            if terminated or truncated:
                done_at = s_ + 1
        push('n_done', done_at)
        for key, vals in ep.items():
            push(key, np.stack(vals))
    out = {k: np.stack(v) for k, v in rec.items()}
    out['n_obs'] = np.array(env.observation_space.shape[0])
    out['obs_low'], out['obs_high'] = env.observation_space.low, env.observation_space.high
    out['n_act'] = np.array(env.action_space.shape[0])
    out['n_bus'] = np.array(len(env.net.bus))
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print(f'{name}: {n} episodes x {S} steps, reward {out["reward"].round(4).tolist()}, '
          f'term {out["terminated"].tolist()} trunc {out["truncated"].tolist()}')


def run(name):
    if name in EPISODE_STEPS:
        return run_episodes(name)
    cls, kwargs, n, seed = SCENARIOS[name]
    env = REF[cls](seed=seed, **kwargs)
    rng = np.random.default_rng(1000 + seed)
    pool = env.train_steps
    rec = {}

    def push(key, val):
        rec.setdefault(key, []).append(np.array(val, copy=True))
    k = -1
    done = 0
    # fixtures with a quota of all-valid rows (scenarios.VALID_ROWS): candidates are recorded until both kinds are full
    want_valid, levels = VALID_ROWS.get(name, (None, ()))
    have = {True: 0, False: 0}
    # quota of rows with a FAILED contingency (scenarios.FAILED_CONTINGENCY_ROWS): the power flows that raise inside the
    # reference's step are counted at the stub's `runpp`
    import pandapower as pp_stub
    want_failed, have_failed, n_raised = FAILED_CONTINGENCY_ROWS.get(name, 0), 0, [0]
    stub_runpp = pp_stub.runpp

    def counting_runpp(*a, **kw_):
        try:
            return stub_runpp(*a, **kw_)
        except pp_stub.powerflow.LoadflowNotConverged:
            n_raised[0] += 1
            raise
    pp_stub.runpp = counting_runpp
    first_steps = list(CANDIDATE_STEPS.get(name, ()))
    while done < n:
        k += 1
        step = int(first_steps.pop(0)) if first_steps else int(rng.choice(pool))
        found = None
        if want_valid is not None and have[True] < want_valid and (k % 2 == 1 or have[False] >= n - want_valid):
            # look for an all-valid state along one scalar action level (see scenarios.VALID_ROWS); every trial is the
            # reference's own reset + step with this candidate's seed, so the recorded draws are those of the last reset
            u = rng.random(env.action_space.shape[0])
            for level in levels:
                env.reset(seed=seed * 100 + k, options={'step': step})
                a = np.clip(level + 0.1 * (u - 0.5), 0.0, 1.0)
                info = env.step(a)[4]
                if 'cost' in info and bool(np.all(info['valids'])):
                    found = a
                    break
            if found is None:
                continue
        obs0, _ = env.reset(seed=seed * 100 + k, options={'step': step})
        log = env.np_random.log
        uni = [u.ravel() for kind, u in log if kind == 'uniform']
        noise = [u.ravel() for kind, u in log if kind == 'random']
        interp = [u.ravel() for kind, u in log if kind == 'random_scalar']
        normal = [u.ravel() for kind, u in log if kind == 'normal']
        normal_noise = (kwargs.get('sampling_params') or {}).get('noise_distribution') == 'normal'
        if normal_noise:                # the noise of _set_simbench_state is drawn with normal()
            noise, normal = normal, []
        action = rng.random(env.action_space.shape[0]) if found is None else found
        if k == 0:
            action = np.clip(action * 1.6 - 0.3, -0.2, 1.2)        # exercise the [0,1] clipping
        snap = snapshot(env.net)
        init_obj = np.sum(env.initial_obj) if env.pf_for_obs else None
        n_raised[0] = 0
        obs, reward, terminated, truncated, info = env.step(action)
        failed_contingencies = n_raised[0] if 'cost' in info else 0
        if want_failed and 'cost' in info:
            if failed_contingencies == 0 and n - done <= want_failed - have_failed:
                continue                                           # the rows still to record must be ones with a failure
            have_failed += failed_contingencies > 0
        if 'cost' not in info:
            # power flow failed (opf_env.py:390-399): keep the inputs as a failure case
            push('fail_step', step)
            push('fail_action', action)
            push('fail_uniform', np.concatenate(uni) if uni else np.zeros(0))
            push('fail_noise', np.concatenate(noise) if noise else np.zeros(0))
            push('fail_interp', np.concatenate(interp) if interp else np.zeros(0))
            push('fail_normal', np.concatenate(normal) if normal else np.zeros(0))
            continue
        if want_valid is not None:
            kind = bool(np.all(info['valids']))
            if have[kind] >= (want_valid if kind else n - want_valid):
                continue                                           # this kind is full: the candidate is not recorded
            have[kind] += 1
        done += 1
        push('step', step)
        push('uniform', np.concatenate(uni) if uni else np.zeros(0))
        push('noise', np.concatenate(noise) if noise else np.zeros(0))
        push('interp', np.concatenate(interp) if interp else np.zeros(0))
        push('normal', np.concatenate(normal) if normal else np.zeros(0))
        push('obs_reset', obs0)
        for key, val in snap.items():
            push(key, val)
        if init_obj is not None:
            push('initial_obj', init_obj)
        push('action', action)
        push('obs_step', obs)
        push('reward', reward)
        push('terminated', terminated)
        push('truncated', truncated)
        push('valids', info['valids'])
        push('violations', info['violations'])
        push('penalties', info['unscaled_penalties'])
        push('cost', info['cost'])
        if want_failed:
            push('failed_contingencies', failed_contingencies)
        push('objective_vector', env.calculate_objective(diff_objective=False))
        push('vm_pu', env.net.res_bus.vm_pu.to_numpy())
        push('va_degree', env.net.res_bus.va_degree.to_numpy())
        push('line_loading', env.net.res_line.loading_percent.to_numpy())
        push('trafo_loading', env.net.res_trafo.loading_percent.to_numpy())
        push('p_ext', env.net.res_ext_grid.p_mw.to_numpy())
        push('q_ext', env.net.res_ext_grid.q_mvar.to_numpy())
        if len(env.net.gen):
            push('q_gen', env.net.res_gen.q_mvar.to_numpy())
        try:
            push('current_actions', env.get_current_actions())
        except KeyError:      # no res_switch / res_trafo.tap_pos in pandapower: only the table route works
            push('current_actions', env.get_current_actions(from_results_table=False))
        for tbl, col in (('sgen', 'q_mvar'), ('sgen', 'p_mw'), ('storage', 'q_mvar'), ('gen', 'p_mw'),
                         ('switch', 'closed'), ('trafo', 'tap_pos'), ('shunt', 'step')):
            if tbl in env.net and len(env.net[tbl]) and col in env.net[tbl].columns:
                push(f'post__{tbl}__{col}', np.array(env.net[tbl][col].to_numpy(dtype=float), copy=True))
    pp_stub.runpp = stub_runpp
    out = {k: np.stack(v) for k, v in rec.items()}
    out['n_obs'] = np.array(env.observation_space.shape[0])
    out['obs_low'], out['obs_high'] = env.observation_space.low, env.observation_space.high
    out['n_act'] = np.array(env.action_space.shape[0])
    out['n_bus'] = np.array(len(env.net.bus))
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    if want_failed:
        print(f'{name}: failed contingencies per sample {out["failed_contingencies"].tolist()}, base-case failures '
              f'{len(out["fail_step"]) if "fail_step" in out else 0}')
    print(f'{name}: {n} samples, obs {out["obs_step"].shape}, reward {out["reward"].round(4)}, '
          f'valid {out["valids"].all(axis=1)}')


def run_e12(name):
    """E12: the reference's own `estimate_reward_distribution` (reward.py:181-216) on a reference environment,
    with every random input it consumes recorded — the step and draws of each reset, each sampled action —
    next to the twelve statistics it returns."""
    import opfgym.reward as ref_reward
    scenario, n = E12_SCENARIOS[name]
    cls, kwargs, _, seed = SCENARIOS[scenario]
    env = REF[cls](seed=seed, **kwargs)
    # `estimate_reward_distribution` calls `env.reset()` without a seed (reward.py:186), which would start an UNSEEDED
    # generator: one seeded reset first, so that this fixture can be regenerated bit for bit
    env.reset(seed=seed)
    rec = {k: [] for k in ('step', 'uniform', 'noise', 'action')}
    ref_reset, ref_sample = env.reset, env.action_space.sample

    def reset(*a, **k):
        env.np_random.log.clear() if hasattr(env, 'np_random') else None
        out = ref_reset(*a, **k)
        log = env.np_random.log
        rec['step'].append(int(env.current_simbench_step))
        rec['uniform'].append(np.concatenate([u.ravel() for kind, u in log if kind == 'uniform'] or [np.zeros(0)]))
        rec['noise'].append(np.concatenate([u.ravel() for kind, u in log if kind == 'random'] or [np.zeros(0)]))
        return out

    def sample():
        a = ref_sample()
        rec['action'].append(np.array(a, copy=True))
        return a
    env.reset, env.action_space.sample = reset, sample
    stats = ref_reward.estimate_reward_distribution(env, num_samples=n)
    out = {k: np.stack(v) for k, v in rec.items()}
    for k, v in stats.items():
        out['stat__' + k] = np.array(float(v))
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print(f'{name}: {n} samples, ' + ', '.join(f'{k}={float(v):.4g}' for k, v in stats.items()))


if __name__ == '__main__':
    names = sys.argv[1:] or (list(SCENARIOS) + list(E12_SCENARIOS))
    for nm in names:
        run_e12(nm) if nm in E12_SCENARIOS else run(nm)
