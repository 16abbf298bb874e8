"""CPU: the environment oracle (oracle/env_oracle.py) reproduces the golden
vectors generated from the reference's own environment classes
(tests/golden/make_golden.py).  Tolerances: table values and observations
1e-12 absolute (same float operations in a different order at most), rewards /
penalties 1e-9, power-flow results 1e-9 (both sides use the same oracle solver).
"""
import os

import numpy as np
import pytest

from env_cases import EPISODE_STEPS, SINGLE_STEP, draws, golden, noise_factors, oracle_env, product_env
from scenarios import E12_SCENARIOS

TAB_TOL = 1e-12
CPU_SAMPLES = {'sc_vc_hv_urban': 3}


@pytest.mark.parametrize('name', SINGLE_STEP)
def test_oracle_matches_reference_golden(name):
    g = golden(name)
    env = product_env(name, defer_device=True)
    assert len(env.net.bus) == int(g['n_bus'])
    assert env.n_actions == int(g['n_act'])
    orc = oracle_env(name, env)
    # (the N-1 fixtures at BASELINE size cost 251 oracle power flows per sample: the CPU suite replays the first three of
    #  `sc_vc_hv_urban`'s six — the GPU suite replays all of them, test_gpu_env.py::test_env_matches_reference_golden)
    n = min(len(g['step']), CPU_SAMPLES.get(name, 10 ** 9))
    for k in range(n):
        noise = noise_factors(name, g['noise'][k])
        d = draws(g, k)
        obs0 = orc.reset(int(g['step'][k]), g['uniform'][k], noise, interp=d['interp'],
                         normal=d['normal'] if d['normal'] is not None else ())
        for key in g:
            if key.startswith('tab__'):
                _, tbl, col = key.split('__')
                assert np.allclose(orc.net[tbl][col].to_numpy(float), g[key][k], rtol=0, atol=TAB_TOL), key
        assert obs0.shape == g['obs_reset'][k].shape
        assert np.allclose(obs0, g['obs_reset'][k], rtol=0, atol=1e-9)
        out = orc.step(g['action'][k])
        assert out['converged']
        assert np.allclose(out['obs'], g['obs_step'][k], rtol=0, atol=1e-9)
        assert np.isclose(out['reward'], g['reward'][k], rtol=0, atol=1e-9)
        assert (out['valids'] == g['valids'][k]).all()
        assert np.allclose(out['violations'], g['violations'][k], rtol=0, atol=1e-9)
        assert np.allclose(out['penalties'], g['penalties'][k], rtol=0, atol=1e-9)
        assert np.isclose(out['cost'], g['cost'][k], rtol=0, atol=1e-9)
        if not orc.n1:
            # (with N-1 the golden objective vector was read after the step, when net.res_*
            # hold the LAST contingency's results — defect D7 — so it is not the base-case one)
            assert np.isclose(out['objective'] + (orc.initial_obj if orc.diff_objective else 0.0),
                              g['objective_vector'][k].sum(), rtol=0, atol=1e-9)
        assert bool(out['terminated']) == bool(g['terminated'][k])
        for key in ('vm_pu', 'va_degree', 'line_loading', 'trafo_loading', 'p_ext', 'q_ext'):
            assert np.allclose(out[key], g[key][k], rtol=0, atol=1e-9, equal_nan=True), key
    if 'fail_step' in g:
        for k in range(len(g['fail_step'])):
            noise = noise_factors(name, g['fail_noise'][k])
            d = draws(g, k, 'fail_')
            orc.reset(int(g['fail_step'][k]), g['fail_uniform'][k], noise, interp=d['interp'],
                      normal=d['normal'] if d['normal'] is not None else ())
            assert not orc.step(g['fail_action'][k])['converged']


@pytest.mark.parametrize('name', list(EPISODE_STEPS))
def test_oracle_multi_step_episodes(name):
    """steps_per_episode > 1: incremental actions (opf_env.py:406-414, 451-458) and
    multi-stage episodes over consecutive time steps (multi_stage.py:26-58)."""
    g = golden(name)
    orc = oracle_env(name)
    for k in range(len(g['step'])):
        obs0 = orc.reset(int(g['step'][k]), g['uniform'][k])
        assert np.allclose(obs0, g['obs_reset'][k], rtol=0, atol=1e-9)
        n_done = int(g['n_done'][k]) if 'n_done' in g else EPISODE_STEPS[name]
        for s_ in range(n_done):
            out = orc.step(g['action'][k, s_])
            assert out['converged']
            assert np.allclose(out['obs'], g['obs_step'][k, s_], rtol=0, atol=1e-9)
            assert np.isclose(out['reward'], g['reward'][k, s_], rtol=0, atol=1e-9)
            assert bool(out['terminated']) == bool(g['terminated'][k, s_])
            assert bool(out['truncated']) == bool(g['truncated'][k, s_])
            assert np.allclose(out['penalties'], g['penalties'][k, s_], rtol=0, atol=1e-9)
            assert np.allclose(out['vm_pu'], g['vm_pu'][k, s_], rtol=0, atol=1e-9)


@pytest.mark.parametrize('name', SINGLE_STEP + list(EPISODE_STEPS))
def test_observation_space_matches_reference(name):
    """get_obs_and_state_space (opf_env.py:720-803) of the host mirror vs the reference's."""
    g = golden(name)
    if 'obs_low' not in g:
        pytest.skip('fixture without space bounds')
    env = product_env(name, defer_device=True)
    assert env.observation_space.shape == g['obs_low'].shape
    assert np.allclose(env.observation_space.low, g['obs_low'], rtol=0, atol=1e-12)
    assert np.allclose(env.observation_space.high, g['obs_high'], rtol=0, atol=1e-12)
    assert env.action_space.shape == (int(g['n_act']),)


@pytest.mark.parametrize('name', list(E12_SCENARIOS))
def test_reward_distribution_statistics_match_the_reference(name):
    """E12: the twelve statistics the reference's own `estimate_reward_distribution` returned
    (tests/golden/make_golden.py run_e12) from the recorded resets and actions."""
    from oracle import env_oracle
    g = golden(name)
    scenario = E12_SCENARIOS[name][0]
    orc = oracle_env(scenario)
    noise = np.stack([noise_factors(scenario, g['noise'][k]) for k in range(len(g['step']))]) \
        if g['noise'].shape[1] and noise_factors(scenario, g['noise'][0]) is not None else None
    stats = env_oracle.estimate_reward_distribution(orc, g['step'], g['uniform'] if g['uniform'].shape[1] else None,
                                                    noise, g['action'])
    for k, v in stats.items():
        assert np.isclose(v, float(g['stat__' + k]), rtol=1e-9, atol=1e-9), k


def test_truncated_normal_transform_has_the_reference_distribution():
    """opf_env.py:304-307 draws with `stats.truncnorm.rvs(min, max, mean, std*diff)` from scipy's own generator,
    which cannot be replayed; the oracle (and the reset kernel) apply the inverse CDF to uniform draws.
    Statistical check: per column, the transformed uniforms are indistinguishable from scipy's `rvs` with the
    same four arguments (two-sample Kolmogorov-Smirnov) and stay inside mean + scale*[min, max]."""
    from scipy import stats
    from oracle import env_oracle
    env = product_env('vc_normal_mean', defer_device=True)
    net = env.net
    rng = np.random.default_rng(0)
    n = 4000
    unit, col, idxs = next(k for k in env.state_keys if 'res_' not in k[0] and 'poly_cost' not in k[0])
    df = net[unit].loc[idxs]
    hi = (df[f'max_max_{col}'] / df.scaling).to_numpy(float)
    lo = (df[f'min_min_{col}'] / df.scaling).to_numpy(float)
    scale = 0.3 * (hi - lo) * (hi - lo)
    mean = df[f'mean_{col}'].to_numpy(float)
    got = np.empty((n, len(idxs)))
    import copy
    work = copy.deepcopy(net)
    for k in range(n):
        env_oracle.sample_truncated_normal(work, [(unit, col, idxs)], iter(rng.random(len(idxs))), 0.3)
        got[k] = work[unit].loc[idxs, col].to_numpy(float)
    for j in range(0, len(idxs), max(1, len(idxs) // 6)):
        if scale[j] == 0:
            continue
        ref = stats.truncnorm.rvs(lo[j], hi[j], mean[j], scale[j], n, random_state=np.random.default_rng(j))
        assert stats.ks_2samp(got[:, j], ref).pvalue > 1e-3
        assert (got[:, j] >= mean[j] + scale[j] * lo[j] - 1e-12).all() and (got[:, j] <= mean[j] + scale[j] * hi[j] + 1e-12).all()


def test_oracle_counts_the_iterations_of_every_solve():
    """`solve_iterations`: one entry per successful power flow of the step — the base case and each contingency of
    security_constrained.py:44-66 (the GPU test of `contingency_start='flat'` compares its sum)."""
    from env_cases import oracle_env, product_env
    host = product_env('sc_hv_small', defer_device=True)
    orc = oracle_env('sc_hv_small', host)
    orc.reset(int(host.train_steps[3]))
    assert len(orc.solve_iterations) <= 1                       # reset: at most the base case (opf_env.py:209-216)
    ref = orc.step(np.full(host.n_actions, 0.5))
    n_cont = sum(len(i) for _, _, i in host.n_minus_one_keys)
    assert ref['converged'] and len(orc.solve_iterations) == 1 + n_cont
    assert all(1 <= it <= 10 for it in orc.solve_iterations)


@pytest.mark.skipif(not os.path.isdir('/root/reference/opfgym'), reason='needs /root/reference (build container only)')
def test_recorded_golden_fixtures_are_reproduced_from_the_reference(tmp_path):
    """The committed generator regenerates its fixtures bit for bit from the reference's own classes — the three
    `estimate_reward_distribution` fixtures included (their first reset is seeded) and a slice of the step scenarios
    (the whole set takes minutes; VERDICT r02 regenerated others by hand)."""
    import subprocess
    import sys
    names = ['e12_vc_mv_small', 'e12_sc_hv_small', 'e12_vc_noisy', 'vc_mv_small', 'qm_mv_small', 'eco_hv_small',
             'maxren_lv', 'vc_mv_3w', 'sc_hv_small']
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1', OPFX_GOLDEN_OUT=str(tmp_path))
    r = subprocess.run([sys.executable, '-B', os.path.join(golden, 'make_golden.py')] + names, env=env, capture_output=True,
                       text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    for n in names:
        a, b = np.load(tmp_path / f'{n}.npz', allow_pickle=False), np.load(os.path.join(golden, f'{n}.npz'), allow_pickle=False)
        assert set(a.files) == set(b.files), n
        for k in a.files:
            same = np.array_equal(a[k], b[k], equal_nan=True) if a[k].dtype.kind == 'f' else np.array_equal(a[k], b[k])
            assert same, (n, k)
