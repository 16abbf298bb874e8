"""Builders shared by the environment parity tests (test code only)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
from scenarios import EPISODE_STEPS, PRODUCT_KWARGS, SCENARIOS  # noqa: E402

SINGLE_STEP = [n for n in SCENARIOS if n not in EPISODE_STEPS]

from opfgym_amd import envs as product_envs  # noqa: E402
from oracle import env_oracle  # noqa: E402


def golden(name):
    return dict(np.load(os.path.join(HERE, 'golden', name + '.npz')))


def product_env(name, batch_size=1, defer_device=False, **extra):
    """`<scenario>+beyond`: the scenario's problem definition on its grid WITH wards, motors, series impedances and a bus-bus
    switch with an impedance added (tests/beyond_simbench.py): static parts of the grid, the definition is otherwise the
    scenario's."""
    base, _, variant = name.partition('+')
    cls, kwargs, _, seed = SCENARIOS[base]
    kw = dict(kwargs)
    kw.update(PRODUCT_KWARGS.get(base, {}))
    kw.update(extra)
    if variant in ('beyond', 'narrowq') and 'definition' not in kw:
        d = getattr(product_envs, cls)(seed=seed, batch_size=1, defer_device=True, **kw).definition
        if variant == 'beyond':
            import beyond_simbench
            beyond_simbench.add_elements(d.net)
        else:
            # `<scenario>+narrowq`: the grid's own generators with reactive ranges narrow enough to bind in most states (the
            # q-limit loop together with whatever the scenario actuates: taps, switches, contingencies)
            n = len(d.net.gen)
            assert n, name
            d.net.gen['min_q_mvar'] = -(2.0 + 1.5 * np.arange(n))
            d.net.gen['max_q_mvar'] = 1.0 + 2.0 * np.arange(n)[::-1]
        kw['definition'] = d
    return getattr(product_envs, cls)(seed=seed, batch_size=batch_size, defer_device=defer_device, **kw)


def reward_dict(rf):
    kind = type(rf).__name__.lower()
    return dict(kind=kind, penalty_weight=rf.penalty_weight, clip_range=rf.clip_range,
                scaling_params=rf.scaling_params, valid_reward=rf.valid_reward,
                invalid_penalty=rf.invalid_penalty, invalid_objective_share=rf.invalid_objective_share)


def oracle_env(name, env=None):
    """EnvOracle for a scenario, built from the host-side problem definition."""
    cls, kwargs, _, _ = SCENARIOS[name.partition('+')[0]]
    env = env or product_env(name, defer_device=True)
    d = env.host_definition()
    tail = env_oracle.TAILS.get(cls)
    if cls in ('VoltageControl', 'QMarket', 'SecurityConstrainedVoltageControl'):          # constructor parameters of the `_sampling` tails
        tail = lambda net, dr: env_oracle.tail_voltage_control(net, dr, bool(env.market_based))
    elif cls == 'LoadShedding':
        tail = lambda net, dr: env_oracle.tail_load_shedding(net, dr, env.storage_efficiency)
    return env_oracle.EnvOracle(
        d['net'], d['act_keys'], d['obs_keys'], d['profiles'], d['constraints'],
        reward_dict(d['reward_function']), tail,
        autoscale_actions=env.autoscale_actions, diff_action_step_size=env.diff_action_step_size,
        clipped_action_penalty=env.clipped_action_penalty, diff_objective=env.diff_objective,
        add_mean_obs=env.add_mean_obs, pf_for_obs=env.pf_for_obs,
        steps_per_episode=env.steps_per_episode, n_minus_one_keys=env.n_minus_one_keys,
        not_converged_penalty=env.not_converged_penalty, data=env.train_data, state_keys=env.state_keys,
        sampling_params=env.sampling_params, bus_wise_obs=env.bus_wise_obs,
        multi_stage=cls == 'MultiStageOpf', split=(env.test_steps, env.validation_steps, env.train_steps),
        objective=(lambda net: np.concatenate([f(net) for f in env.objective_terms])) if env.objective_terms else None)


def noise_factors(name, raw):
    """Recorded raw U[0,1) draws -> multiplicative factors (opf_env.py:354-355);
    None when the scenario samples without noise (factor exactly 1.0)."""
    kwargs = SCENARIOS[name][1]
    sp = kwargs.get('sampling_params') or {}
    nf = sp.get('noise_factor', 0.1 if kwargs.get('train_data') == 'mixed' else 0.0)   # opf_env.py:318 default
    if not nf or raw.size == 0:
        return None
    if sp.get('noise_distribution') == 'normal':
        return raw                       # standard-normal draws; the factor is applied by the sampler
    return raw * nf * 2 + (1 - nf)


def draws(g, k, prefix=''):
    """interp / normal draw vectors of sample k (absent in older fixtures)."""
    out = {}
    for key in ('interp', 'normal'):
        arr = g.get(prefix + key)
        out[key] = arr[k] if arr is not None and arr.shape[1] else None
    return out


def mixed_modes(name, g, prefix=''):
    """'mixed' sampling: the reset's first draw r (recorded like an interpolation draw) -> data source
    per sample (opf_env.py:244-251); None for other scenarios."""
    kwargs = SCENARIOS[name][1]
    if kwargs.get('train_data') != 'mixed':
        return None
    p = (kwargs.get('sampling_params') or {}).get('data_probabilities', (0.5, 0.75, 1.0))
    r = g[prefix + 'interp'][:, 0]
    return (r >= p[0]).astype(np.int32) + (r >= p[1]).astype(np.int32)


def _oracle_rows_worker(args):
    """(worker process) reset + step of the oracle environment for a share of the rows: [(row, obs at reset, step output)]."""
    name, items = args
    orc = oracle_env(name)
    out = []
    for k, step, uni, act in items:
        ob0 = orc.reset(int(step), uni if uni is not None else ())
        out.append((k, ob0, orc.step(act)))
    return out


def oracle_rows_parallel(name, steps, uniform, actions, rows, n_proc=8):
    """The oracle's reset + step of `rows` in `n_proc` worker processes (spawned: the caller may hold a GPU context) — the
    N-1 configuration costs ~3 s of oracle per row.  Returns {row: (obs at reset, step output)}."""
    import multiprocessing as mp
    rows = [int(k) for k in rows]
    shares = [rows[w::n_proc] for w in range(n_proc)]
    jobs = [(name, [(k, steps[k], None if uniform is None else uniform[k], actions[k]) for k in share]) for share in shares if share]
    with mp.get_context('spawn').Pool(len(jobs)) as pool:
        parts = pool.map(_oracle_rows_worker, jobs)
    return {k: (ob0, ref) for part in parts for k, ob0, ref in part}
