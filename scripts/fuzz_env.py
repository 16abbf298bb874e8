"""Differential fuzz of the fused environment path against the CPU oracle (developer script, GPU box):
random combinations of the in-scope OpfEnv options on the small scenario grids, one reset
and up to three steps each, every instance compared with the oracle on the same draws.

    python scripts/fuzz_env.py [n_configs] [seed] [only_config]        (OPFX_FUZZ_BASES=a,b: draw from these scenarios only;
                                                                        OPFX_FUZZ_INIT=auto|dc, OPFX_FUZZ_FAITHFUL=1: see run_one)

Prints one line per configuration and a summary; exits non-zero on the first mismatch."""
import os
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
from env_cases import oracle_env, product_env  # noqa: E402
from opfgym_amd import capi  # noqa: E402
capi.set_default_debug(capi.debug_from_env())      # this harness is steered through OPFX_* variables (see capi.debug_from_env)

R_TOL, V_TOL = 1e-6, 1e-7      # (both Newton solvers stop at ||F||inf < 1e-8 p.u.; north-star bar 1e-6 p.u.)
REL = 1e-6          # relative part for rewards/penalties (penalty_power 2 and large cost coefficients amplify the 1e-8 p.u. solver tolerance)
BASES = ['vc_mv_small', 'qm_mv_small', 'eco_hv_small', 'eco_hv_small_shared', 'maxren_lv', 'loadshed_mv_small', 'mixed_lv',
         'sc_hv_small', 'reconf_hv_small_sw', 'shunt_hv_small_sw', 'busbar_hv_small_sw', 'nonsimbench_case9', 'constraint_sat_lv', 'partial_obs_lv',
         'custom_constraint_lv', 'multistage_lv', 'vc_mv_3w']


def pick(rng, seq):
    return seq[int(rng.integers(len(seq)))]


def random_options(rng):
    kw = {}
    rf = pick(rng, ['summation', 'replacement', 'parameterized'])
    kw['reward_function'] = rf
    rp = {}
    if rng.random() < 0.5:
        rp['penalty_weight'] = pick(rng, [None, 0.2, 0.7])
    if rng.random() < 0.3:
        rp['clip_range'] = (-2.0, 1.0)
    if rf == 'replacement' and rng.random() < 0.5:
        rp['valid_reward'] = 0.7
    if rf == 'parameterized':
        rp.update(valid_reward=pick(rng, [0.0, 1.0]), invalid_penalty=pick(rng, [0.0, 0.5]),
                  invalid_objective_share=pick(rng, [0.0, 0.3, 1.0]))
    if rp:
        kw['reward_function_params'] = rp
    for flag, p in (('diff_objective', 0.3), ('add_res_obs', 0.4), ('add_act_obs', 0.4), ('add_mean_obs', 0.3)):
        if rng.random() < p:
            kw[flag] = True
    if rng.random() < 0.25:
        kw['autoscale_actions'] = False              # (needs min_min_/max_max_ columns, as in the reference)
    spe = pick(rng, [1, 1, 3])
    kw['steps_per_episode'] = spe
    if spe > 1 and rng.random() < 0.6:
        kw['diff_action_step_size'] = pick(rng, [0.1, 0.3])
    if rng.random() < 0.3:
        kw['clipped_action_penalty'] = 0.5
    if rng.random() < 0.3:
        kw['initial_action'] = 'random'
    cp = {}
    if rng.random() < 0.3:
        cp['only_worst_case_violations'] = True
    if rng.random() < 0.3:
        cp['penalty_power'] = pick(rng, [0.5, 2.0])
    if rng.random() < 0.3:
        cp['penalty_factor'] = 3.0
    if rng.random() < 0.3:
        cp['violation_count_penalty'] = 0.1
    if rng.random() < 0.2:
        cp['autoscale_violation'] = False
    if cp:
        kw['constraint_params'] = cp
    data = pick(rng, ['simbench', 'simbench', 'noisy_simbench', 'full_uniform', 'normal_around_mean', 'mixed'])
    kw['train_data'] = data
    if data == 'mixed':
        kw['test_data'] = pick(rng, ['mixed', 'simbench', 'full_uniform'])
        if rng.random() < 0.5:
            kw['sampling_params'] = dict(data_probabilities=(0.3, 0.6, 1.0))
        return kw
    sp = {}
    if data in ('simbench', 'noisy_simbench'):
        if rng.random() < 0.4:
            sp['noise_factor'] = pick(rng, [0.05, 0.1])
        if sp and rng.random() < 0.4:
            sp['noise_distribution'] = 'normal'
        if rng.random() < 0.3:
            sp['interpolate_steps'] = True
    if data == 'normal_around_mean' and rng.random() < 0.5:
        sp['relative_std'] = 0.1
    if sp:
        kw['sampling_params'] = sp
    if rng.random() < 0.3:
        kw['test_data'] = pick(rng, ['simbench', 'noisy_simbench', 'full_uniform'])
    return kw


def class_options(rng, base):
    """Constructor parameters of the benchmark classes (they change the grid the env is built on)."""
    kw = {}
    base = base.partition('+')[0]          # (`<scenario>+beyond`: the scenario with wards, motors, impedances ... added, env_cases.product_env)
    if base not in ('nonsimbench_case9', 'constraint_sat_lv') and rng.random() < 0.5:
        kw['grid_seed'] = int(rng.integers(1, 50))      # another instance of the synthetic grid family
    if base in ('vc_mv_small', 'qm_mv_small', 'vc_mv_3w'):
        if rng.random() < 0.5:
            kw['load_scaling'] = pick(rng, [1.2, 2.0])
        if rng.random() < 0.5:
            kw['gen_scaling'] = pick(rng, [1.0, 1.6])
        if rng.random() < 0.4:
            kw['cos_phi'] = 0.9
        if rng.random() < 0.4:
            kw['max_q_exchange'] = pick(rng, [0.2, 1.0])
        if rng.random() < 0.3:
            kw['market_based'] = base == 'vc_mv_small'
        if base == 'vc_mv_small' and rng.random() < 0.25:
            kw['bus_wise_obs'] = True
    elif base == 'eco_hv_small':
        if rng.random() < 0.3:
            kw['simbench_network_name'] = '1-HV-mixed--0-sw'     # 306 buses: the wave-team kernel
            kw.pop('grid_seed', None)
        if rng.random() < 0.5:
            kw['max_price_eur_gwh'] = pick(rng, [0.3, 1.0])
        if rng.random() < 0.4:
            kw['load_scaling'] = 1.2
    elif base == 'loadshed_mv_small':
        if rng.random() < 0.5:
            kw['max_p_exchange'] = pick(rng, [4.0, 12.0])
        if rng.random() < 0.4:
            kw['storage_efficiency'] = 0.9
    elif base == 'sc_hv_small':
        if rng.random() < 0.3:
            # 372 buses: the wave-team kernel, N-1; OPFX_FUZZ_SC_GRID=1-HV-mixed--0-sw: the 306-bus grid instead — three teams per CU on
            # the plan with shared LDS slots, compiled for three wavefronts per SIMD, WITH contingency modifiers
            kw['simbench_network_name'] = os.environ.get('OPFX_FUZZ_SC_GRID', '1-HV-urban--0-sw')
            kw.pop('grid_seed', None)
        if rng.random() < 0.5:
            kw['n_minus_one_lines'] = pick(rng, [(1,), (0, 2, 5), (1, 3, 7, 9)])
        if rng.random() < 0.3:
            kw['not_converged_penalty'] = 5.0
    return kw


def np_(t):
    return t.detach().cpu().numpy()


def run_one(base, kw, rng, B=8):
    # OPFX_FUZZ_INIT=auto|dc: the product with pandapower's start (a DC power flow first where the grid is fed above 70 kV):
    # the DC-start kernels and their specialisations against the same oracle — same fixed point, same converged set away from
    # voltage collapse (nothing is drawn for it, so a campaign replays with and without)
    pk = dict(kw, init=os.environ['OPFX_FUZZ_INIT']) if os.environ.get('OPFX_FUZZ_INIT') else kw
    # OPFX_FUZZ_FAITHFUL=1: the product under the reference's own solver settings (reference_faithful=True: DC start, every N-1
    # contingency from scratch — from round 6 on from a rank-1 update of the base case's DC angles —, pypower's q-limit path)
    if os.environ.get('OPFX_FUZZ_FAITHFUL'):
        pk = dict(pk, reference_faithful=True)
    env = product_env(base, batch_size=B, **pk)
    orc = oracle_env(base, product_env(base, defer_device=True, **kw))
    is_test = 'test_data' in kw and rng.random() < 0.5
    distr = env.test_data if is_test else env.train_data
    if env.mixed and distr == 'mixed':
        return run_mixed(env, orc, kw, rng, B)
    steps = rng.choice(env.test_steps if is_test else env.train_steps, B)
    uniform = rng.random((B, env.n_uniform)) if env.n_uniform else None
    noisy = env.n_noise and (distr == 'noisy_simbench' or 'noise_factor' in env.sampling_params)
    noise = None
    if noisy and env.noise_distribution == 'normal':
        noise = rng.standard_normal((B, env.n_noise))
    elif noisy:
        noise = rng.random((B, env.n_noise)) * (2 * env.noise_factor) + (1 - env.noise_factor)
    normal = rng.standard_normal((B, env.n_normal)) if getattr(env, 'n_normal', 0) else None
    interp = rng.random((B, len(env.tables))) if env.interpolate_steps and env.uses_profiles else None
    opts = {'step': steps, 'uniform': uniform, 'noise': noise, 'test': is_test}
    if normal is not None:
        opts['normal'] = normal
    if interp is not None:
        opts['interp'] = interp
    if kw.get('initial_action') == 'random':
        opts['initial_action'] = rng.random((B, env.n_actions))
    obs0, _ = env.reset(options=opts)
    obs0 = np_(obs0).copy()
    n_steps = kw.get('steps_per_episode', 1)
    acts = rng.random((n_steps, B, env.n_actions))
    outs = []
    for s_ in range(n_steps):
        o = env.step(acts[s_])
        outs.append(dict(obs=np_(o[0]).copy(), reward=np_(o[1]).copy(), term=np_(o[2]).copy(), trunc=np_(o[3]).copy(),
                         conv=np_(o[4]['converged']).copy(), valids=np_(o[4]['valids']).copy(),
                         viol=np_(o[4]['violations']).copy(), pen=np_(o[4]['unscaled_penalties']).copy(),
                         cost=np_(o[4]['cost']).copy(), vm=np_(env.result_table('bus', 'vm_pu')).copy(),
                         q_gen=np_(env.result_table('gen', 'q_mvar')).copy() if len(env.net.gen) else None))
    checked = 0
    for k in range(B):
        # (the reference draws sequentially and only what the data source needs; the product has one
        # fixed draw column per op)
        src = env.source_of[distr]
        rk = dict(uniform=uniform[k][env.ops.uniform_columns(src)] if uniform is not None else (),
                  noise=noise[k] if noise is not None else None)
        extra = {'data': distr}
        if normal is not None:
            extra['normal'] = normal[k]
        if interp is not None:
            extra['interp'] = interp[k]
        if 'initial_action' in opts:
            extra['initial_action'] = opts['initial_action'][k]
        try:
            ob0 = orc.reset(int(steps[k]), rk['uniform'], rk['noise'], **extra)
        except AssertionError as e:
            if e.args:
                raise
            # the oracle's power flow failed in reset (env_oracle: `assert self.solve()`): the product has
            # re-sampled this row (opf_env.py:209-214), nothing to compare
            assert not np.allclose(obs0[k], 0.0), ('product row after a failed reset', k)
            print(f'      row {k}: reset power flow fails in the oracle at step {int(steps[k])}; re-sampled by the product')
            continue
        assert np.allclose(obs0[k], ob0, rtol=0, atol=R_TOL, equal_nan=True), ('reset obs', k, np.abs(obs0[k] - ob0).max(),
                                                                 'nan product/oracle', int(np.isnan(obs0[k]).sum()), int(np.isnan(ob0).sum()),
                                                                 'first bad', int(np.flatnonzero(~np.isclose(obs0[k], ob0, rtol=0, atol=R_TOL, equal_nan=True))[0]), 'is_test', is_test)
        for s_ in range(n_steps):
            ref = orc.step(acts[s_, k])
            got = outs[s_]
            assert bool(got['conv'][k]) == bool(ref['converged']), ('converged', k, s_)
            if not ref['converged']:
                break
            # (multi_stage.py:56 builds the next stage's observation WITHOUT the mean entries that
            # add_mean_obs appends — the reference's observation changes length there, defect D13; the
            # batched env keeps the fixed shape of its observation space: the common prefix is compared)
            go = got['obs'][k][:len(ref['obs'])] if base == 'multistage_lv' else got['obs'][k]
            assert np.allclose(go, ref['obs'], rtol=0, atol=R_TOL, equal_nan=True), ('obs', k, s_, np.abs(go - ref['obs']).max())
            # (with diff_objective the reward is the difference of two objectives: the solver tolerance enters
            #  relative to THEIR size — a 9-bus case with generator costs of several thousand per hour — not to the
            #  size of the difference)
            big = abs(float(getattr(orc, 'initial_obj', 0.0) or 0.0)) if kw.get('diff_objective') else 0.0
            assert abs(got['reward'][k] - ref['reward']) <= R_TOL + REL * max(abs(ref['reward']), big), \
                ('reward', k, s_, got['reward'][k], ref['reward'])
            nc = len(ref['valids'])
            assert (got['valids'][k][:nc] == ref['valids']).all(), ('valids', k, s_)
            assert np.allclose(got['viol'][k][:nc], ref['violations'], rtol=REL, atol=R_TOL), ('violations', k, s_)
            # (penalty = factor * violation ** power: the relative error of a violation enters `power` times)
            rel_pen = REL * max(1.0, float((kw.get('constraint_params') or {}).get('penalty_power', 1.0)))
            assert np.allclose(got['pen'][k][:nc], ref['penalties'], rtol=rel_pen, atol=R_TOL), ('penalties', k, s_)
            assert np.isclose(got['cost'][k], ref['cost'], rtol=REL, atol=R_TOL), ('cost', k, s_)
            assert bool(got['term'][k]) == bool(ref['terminated']), ('terminated', k, s_)
            if not env.n_minus_one_keys:          # (after an N-1 step the tables hold the last contingency, D7)
                assert np.allclose(got['vm'][k], ref['vm_pu'], rtol=0, atol=V_TOL, equal_nan=True), ('vm', k, s_, float(np.nanmax(np.abs(got['vm'][k] - ref['vm_pu']))))
                if got['q_gen'] is not None and 'q_gen' in ref:      # res_gen.q_mvar per generator (pfsoln's split; the generators of DC lines last)
                    assert np.allclose(got['q_gen'][k], ref['q_gen'], rtol=REL, atol=1e-6), ('q_gen', k, s_, float(np.abs(got['q_gen'][k] - ref['q_gen']).max()))
            checked += 1
            if ref['terminated'] or ref.get('truncated'):
                break
    return checked


def run_mixed(env, orc, kw, rng, B):
    """train_data='mixed' (opf_env.py:242-251): the first draw r of a reset picks the data source."""
    steps = rng.choice(env.train_steps, B)
    r = rng.random(B)
    p0, p1 = env.data_probabilities[0], env.data_probabilities[1]
    mode = (r >= p0).astype(np.int32) + (r >= p1).astype(np.int32)
    uniform = rng.random((B, env.n_uniform)) if env.n_uniform else None
    normal = rng.standard_normal((B, env.n_normal)) if env.n_normal else None
    noise = rng.random((B, env.n_noise)) * (2 * env.noise_factor) + (1 - env.noise_factor) if env.n_noise else None
    opts = {'step': steps, 'uniform': uniform, 'normal': normal, 'noise': noise, 'mode': mode}
    init = rng.random((B, env.n_actions)) if kw.get('initial_action') == 'random' else None
    if init is not None:
        opts['initial_action'] = init
    obs0 = np_(env.reset(options=opts)[0]).copy()
    acts = rng.random((B, env.n_actions))
    o = env.step(acts)
    got = dict(obs=np_(o[0]), reward=np_(o[1]), conv=np_(o[4]['converged']), vm=np_(env.result_table('bus', 'vm_pu')))
    checked = 0
    for k in range(B):
        m = int(mode[k])
        uni = uniform[k][env.ops.uniform_columns(m)] if uniform is not None else ()
        try:
            ob0 = orc.reset(int(steps[k]), uni, noise[k] if (noise is not None and m == 0) else None,
                            interp=[r[k]], normal=normal[k] if normal is not None else (),
                            initial_action=init[k] if init is not None else None)
        except AssertionError as e:
            if e.args:
                raise
            continue          # reset power flow fails in the oracle: the product re-sampled this row
        assert np.allclose(obs0[k], ob0, rtol=0, atol=R_TOL, equal_nan=True), ('mixed reset obs', k, m, np.abs(obs0[k] - ob0).max())
        ref = orc.step(acts[k])
        assert bool(got['conv'][k]) == bool(ref['converged']), ('converged', k)
        if not ref['converged']:
            continue
        assert np.allclose(got['obs'][k], ref['obs'], rtol=0, atol=R_TOL, equal_nan=True), ('obs', k)
        # (as in run_one: with diff_objective the solver tolerance enters relative to the size of the two objectives; a penalty
        #  with penalty_power < 1 — the square root of a violation — amplifies it for violations near zero: d sqrt(v) = dv / (2 sqrt(v)))
        big = abs(float(getattr(orc, 'initial_obj', 0.0) or 0.0)) if kw.get('diff_objective') else 0.0
        soft = 10.0 if float((kw.get('constraint_params') or {}).get('penalty_power', 1.0)) < 1.0 else 1.0
        assert abs(got['reward'][k] - ref['reward']) <= soft * R_TOL + REL * max(abs(ref['reward']), big), ('reward', k, got['reward'][k], ref['reward'])
        if not env.n_minus_one_keys:
            assert np.allclose(got['vm'][k], ref['vm_pu'], rtol=0, atol=V_TOL, equal_nan=True), ('vm', k)
        checked += 1
    return checked


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    only = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].lstrip('-').isdigit() else None      # replay one configuration
    if os.environ.get('OPFX_FUZZ_BASES'):                       # e.g. OPFX_FUZZ_BASES=shunt_hv_small_sw,reconf_hv_small_sw
        BASES[:] = os.environ['OPFX_FUZZ_BASES'].split(',')
    total = bad = 0
    for c in range(n):
        if only is not None and c != only:
            continue
        rng = np.random.default_rng([seed, c])
        base = pick(rng, BASES)
        kw = dict(random_options(rng), **class_options(rng, base))
        if base in ('nonsimbench_case9', 'multistage_lv'):   # (own distributions / re-sampling inside step)
            for key in ('train_data', 'test_data', 'sampling_params'):
                kw.pop(key, None)
        if base == 'multistage_lv':
            kw['steps_per_episode'] = pick(rng, [2, 4])
        try:
            checked = run_one(base, kw, rng)
            total += checked
            print(f'[{c}] ok   {base} checked={checked} {kw}')
        except (NotImplementedError, KeyError, ImportError) as e:
            print(f'[{c}] skip {base} {kw}: {e}')
        except AssertionError as e:
            bad += 1
            print(f'[{c}] MISMATCH {base} {kw}: {e.args}')
        except Exception:
            bad += 1
            print(f'[{c}] ERROR {base} {kw}')
            traceback.print_exc()
    print(f'{n} configurations, {total} instance-steps compared, {bad} failures')
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
