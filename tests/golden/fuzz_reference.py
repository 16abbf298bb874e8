"""Differential fuzz of the CPU oracle's environment half against the REFERENCE's own classes.

Runs only in the build container (needs /root/reference), CPU only:

    PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/fuzz_reference.py [n_configs] [seed] [only_config]

The golden fixtures pin 31 fixed scenarios; this script widens the pin: random combinations of the
in-scope `OpfEnv` options and benchmark-class parameters (the generator of scripts/fuzz_env.py) are
run through the reference's environment classes — imported from /root/reference under the stub
packages of tests/golden/_stubs, every `pp.runpp` answered by oracle/pf_oracle.py, exactly as
make_golden.py does — and through oracle/env_oracle.py on the same recorded random draws.  Nothing
is stored; a mismatch points at a misreading that the product and the oracle could otherwise share.
"""
import importlib.util
import os
import sys
import traceback

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(HERE, '_stubs'), ROOT, '/root/reference', HERE, os.path.join(ROOT, 'tests')]

import numpy as np  # noqa: E402

import make_golden as mg  # noqa: E402  (reference classes under stubs)
from scenarios import SCENARIOS  # noqa: E402
import env_cases  # noqa: E402
from opfgym_amd import envs as product_envs  # noqa: E402

_spec = importlib.util.spec_from_file_location('fuzz_env', os.path.join(ROOT, 'scripts', 'fuzz_env.py'))
fz = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(fz)

TOL = 1e-9
BASES = list(fz.BASES)           # every class of make_golden.REF (the five benchmarks and the nine examples)


def noise_factors(kwargs, raw, distr):
    sp = kwargs.get('sampling_params') or {}
    nf = sp.get('noise_factor', 0.1 if distr in ('noisy_simbench', 'mixed') else 0.0)
    if not nf or raw.size == 0:
        return None
    if sp.get('noise_distribution') == 'normal':
        return raw
    return raw * nf * 2 + (1 - nf)


def run_one(base, kw, rng, n_samples=3):
    cls, base_kwargs, _, seed = SCENARIOS[base]
    kwargs = dict(base_kwargs)
    kwargs.update(kw)
    kwargs.pop('grid_seed', None)                     # (a parameter of this repo's synthetic grids only)
    kwargs.pop('n_minus_one_lines', None)             # (the reference example hard-codes lines 1, 3, 7)
    ref = mg.REF[cls](seed=seed, **kwargs)
    prod = getattr(product_envs, cls)(seed=seed, batch_size=1, defer_device=True, **kwargs)
    orc = env_cases.oracle_env(base, prod)
    # spaces of the host mirror (get_obs_and_state_space, opf_env.py:720-803)
    assert prod.action_space.shape == ref.action_space.shape, ('action space', prod.action_space.shape, ref.action_space.shape)
    assert prod.observation_space.shape == ref.observation_space.shape, ('obs space shape',)
    assert np.allclose(prod.observation_space.low, ref.observation_space.low, rtol=0, atol=1e-9, equal_nan=True), ('obs low',)
    assert np.allclose(prod.observation_space.high, ref.observation_space.high, rtol=0, atol=1e-9, equal_nan=True), ('obs high',)
    orc.carry_over = True        # the reference keeps its net between episodes (see EnvOracle.carry_over, D12)
    S = kwargs.get('steps_per_episode', 1)
    sampled = []                                      # initial_action='random': the action space's own draws
    space_sample = ref.action_space.sample

    def logged_sample(*a, **k):
        sampled.append(np.array(space_sample(*a, **k), dtype=float))
        return sampled[-1]
    ref.action_space.sample = logged_sample
    checked = 0
    for k in range(n_samples):
        is_test = 'test_data' in kwargs and rng.random() < 0.4
        distr = kwargs.get('test_data', 'simbench') if is_test else kwargs.get('train_data', 'simbench')
        pool = ref.test_steps if is_test else ref.train_steps
        step = int(rng.choice(pool))
        del sampled[:]
        try:
            obs0, _ = ref.reset(seed=seed * 100 + k, options={'step': step, 'test': is_test})
        except RecursionError:
            continue                                   # (the reference re-samples failed resets recursively)
        log = ref.np_random.log
        cat = lambda kind: (np.concatenate([u.ravel() for kd, u in log if kd == kind])
                            if any(kd == kind for kd, _ in log) else np.zeros(0))
        uni, noise, interp, normal = cat('uniform'), cat('random'), cat('random_scalar'), cat('normal')
        if (kwargs.get('sampling_params') or {}).get('noise_distribution') == 'normal' and distr != 'normal_around_mean':
            noise, normal = normal, np.zeros(0)
        if any(kd not in ('uniform', 'random', 'random_scalar', 'normal') for kd, _ in log):
            raise AssertionError(('unexpected draw kinds', sorted({kd for kd, _ in log})))
        ob = orc.reset(step, uni, noise_factors(kwargs, noise, distr), interp=interp if interp.size else None,
                       normal=normal if normal.size else (), data=distr,
                       initial_action=sampled[-1] if sampled else None)
        assert np.allclose(obs0, ob, rtol=0, atol=TOL, equal_nan=True), ('reset obs', k, float(np.nanmax(np.abs(obs0 - ob))))
        for s_ in range(S):
            action = rng.random(ref.action_space.shape[0])
            obs, reward, term, trunc, info = ref.step(action)
            out = orc.step(action)
            assert out['converged'] == ('cost' in info), ('converged', k, s_)
            if not out['converged']:
                break
            assert np.allclose(obs, out['obs'], rtol=0, atol=TOL, equal_nan=True), ('obs', k, s_, float(np.nanmax(np.abs(obs - out['obs']))))
            assert np.isclose(reward, out['reward'], rtol=1e-9, atol=TOL), ('reward', k, s_, float(reward), out['reward'])
            assert (np.asarray(info['valids']) == out['valids']).all(), ('valids', k, s_)
            assert np.allclose(info['violations'], out['violations'], rtol=1e-9, atol=TOL), ('violations', k, s_)
            assert np.allclose(info['unscaled_penalties'], out['penalties'], rtol=1e-9, atol=TOL), ('penalties', k, s_)
            assert np.isclose(info['cost'], out['cost'], rtol=1e-9, atol=TOL), ('cost', k, s_)
            assert bool(term) == bool(out['terminated']) and bool(trunc) == bool(out.get('truncated', False)), ('flags', k, s_)
            checked += 1
            if term or trunc:
                break
    return checked


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    only = int(sys.argv[3]) if len(sys.argv) > 3 else None
    total = bad = 0
    for c in range(n):
        if only is not None and c != only:
            continue
        rng = np.random.default_rng([seed, c])
        base = fz.pick(rng, BASES)
        kw = dict(fz.random_options(rng), **fz.class_options(rng, base))
        kw.pop('simbench_network_name', None)
        if base in ('nonsimbench_case9', 'multistage_lv'):      # (own distributions / re-sampling inside step)
            for key in ('train_data', 'test_data', 'sampling_params'):
                kw.pop(key, None)
        if base == 'multistage_lv':
            kw['steps_per_episode'] = fz.pick(rng, [2, 4])
        try:
            checked = run_one(base, kw, rng)
            total += checked
            print(f'[{c}] ok   {base} checked={checked} {kw}')
        except (NotImplementedError, KeyError, TypeError) as e:
            print(f'[{c}] skip {base} {kw}: {type(e).__name__} {e}')
        except AssertionError as e:
            bad += 1
            print(f'[{c}] MISMATCH {base} {kw}: {e.args}')
        except Exception:
            bad += 1
            print(f'[{c}] ERROR {base} {kw}')
            traceback.print_exc()
    print(f'{n} configurations, {total} instance-steps compared with the reference, {bad} failures')
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
