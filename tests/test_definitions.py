"""CPU: the recorded problem definitions (opfgym_amd/definitions/) — DATA written by
tests/golden/make_definitions.py from the reference's own environment classes — load everywhere, and, where the
reference is at hand (the build container), are reproduced bit for bit by that script and equal what
`BatchedOpfEnv.from_reference` reads off the live reference objects."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from opfgym_amd import definition, envs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HAVE_REFERENCE = os.path.isdir('/root/reference/opfgym')


def test_every_recorded_definition_loads():
    idx = json.load(open(os.path.join(definition.DEF_DIR, 'index.json')))
    assert len(idx) >= 20
    for key, name in idx.items():
        d = definition.load(os.path.join(definition.DEF_DIR, name))
        assert len(d.net.bus) > 0 and len(d.act_keys) > 0 and len(d.obs_keys) > 0, name
        ref_path = json.loads(key)[0]
        assert ref_path.startswith('opfgym.') and ref_path.rsplit('.', 1)[1] == d.class_name
        for unit, col, idxs in d.act_keys + d.obs_keys + d.state_keys:
            assert set(idxs) <= set(d.net[unit].index), (name, unit, col)
        if d.profiles:
            for (unit, col), df in d.profiles.items():
                assert set(df.columns) <= set(d.net[unit].index) and df.shape[0] == 35136, (name, unit, col)


def test_unrecorded_arguments_fail_loudly_without_the_reference():
    if definition.reference_class('opfgym.envs.VoltageControl') is not None:
        pytest.skip('the reference is importable here')
    # (classes with a native recipe — the five benchmark environments — build for any arguments,
    #  tests/test_native_definition.py; the example classes exist as recorded definitions only)
    with pytest.raises(ImportError, match='no recorded definition'):
        envs.MultiStageOpf(simbench_network_name='mv-small', batch_size=1, defer_device=True)


@pytest.mark.skipif(not HAVE_REFERENCE, reason='needs /root/reference (build container only)')
def test_recorded_definitions_are_reproduced_from_the_reference(tmp_path):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1', OPFX_DEF_OUT=str(tmp_path))
    r = subprocess.run([sys.executable, '-B', os.path.join(ROOT, 'tests', 'golden', 'make_definitions.py')],
                       env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    new = json.load(open(tmp_path / 'index.json'))
    old = json.load(open(os.path.join(definition.DEF_DIR, 'index.json')))
    assert new == old
    for name in old.values():
        a, b = np.load(tmp_path / name), np.load(os.path.join(definition.DEF_DIR, name))
        assert set(a.files) == set(b.files), name
        for k in a.files:
            x, y = a[k], b[k]
            same = np.array_equal(x, y, equal_nan=True) if x.dtype.kind == 'f' else np.array_equal(x, y)
            assert same, (name, k)


@pytest.mark.skipif(not HAVE_REFERENCE, reason='needs /root/reference (build container only)')
def test_from_reference_reads_the_same_definition_off_live_reference_objects():
    r = subprocess.run([sys.executable, '-B', os.path.join(ROOT, 'tests', 'golden', 'check_from_reference.py')],
                       env=dict(os.environ, PYTHONDONTWRITEBYTECODE='1'), capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count('same definition') >= 10


def test_bench_checker_leg_reproduces_the_oracle(tmp_path):
    """bench.py's `--cpu-check` child process (the checker behind `config.max_abs_v_err_pu`): the oracle's base-case
    voltages for the instances a GPU evaluation hands over — here compared with running the oracle directly."""
    import json
    import subprocess
    import sys
    import os
    import numpy as np
    sys.path[:0] = [os.path.dirname(os.path.abspath(__file__))]
    from env_cases import oracle_env, product_env
    from oracle import env_oracle
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    host = product_env('maxren_lv', defer_device=True)
    rng = np.random.default_rng(3)
    d = {'scenario': 'maxren_lv', 'steps': rng.choice(host.train_steps, 2).tolist(), 'uniform': None,
         'actions': rng.random((2, host.n_actions)).tolist()}
    path = tmp_path / 'inputs.json'
    path.write_text(json.dumps(d))
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--cpu-check', str(path)], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1]
    vm = np.array(json.loads(line)['vm'], dtype=float)
    orc = oracle_env('maxren_lv', host)
    for k in range(2):
        orc.reset(int(d['steps'][k]))
        env_oracle.apply_actions(orc.net, orc.act_keys, np.asarray(d['actions'][k]), orc.autoscale, orc.diff_step)
        assert orc.solve()
        assert np.allclose(vm[k], orc.net['res_bus']['vm_pu'].to_numpy(float), rtol=0, atol=0)


def test_reference_faithful_is_one_switch_and_explicit_options_win():
    """`reference_faithful=True` = init='auto' (pandapower's default: 'dc' on grids fed above 70 kV),
    contingency_start='flat' (security_constrained.py:53: every contingency a fresh runpp), carry_over_state=True (D12);
    the default is the fast path, which `reference_deviations` names; an explicitly given option always wins."""
    from opfgym_amd import envs
    kw = dict(simbench_network_name='hv-small', batch_size=1, defer_device=True, seed=0)
    fast = envs.EcoDispatch(**kw)
    assert fast.reference_deviations == {'init': 'flat', 'contingency_start': 'base_case', 'carry_over_state': False, 'pin_point_q_ranges': True}
    assert fast.solve_opts.enforce_q_lims == 1
    assert fast.init == 'flat' and fast.solve_opts.contingency_start == 0 and not fast.carry_over_state and not fast.reference_faithful
    ref = envs.EcoDispatch(reference_faithful=True, **kw)
    assert ref.reference_deviations == {} and ref.reference_faithful
    assert ref.init == 'dc' and ref.solve_opts.init == 1 and ref.solve_opts.contingency_start == 1 and ref.carry_over_state
    assert ref.solve_opts.enforce_q_lims == 2 and not ref.pin_point_q_ranges          # (pypower's own q-limit path)
    mixed = envs.EcoDispatch(reference_faithful=True, init='flat', **kw)
    assert mixed.reference_deviations == {'init': 'flat'} and mixed.init == 'flat' and mixed.solve_opts.contingency_start == 1
    # below 70 kV pandapower's 'auto' IS the flat start
    lv = envs.MaxRenewable(simbench_network_name='1-LV-rural1--0-sw', min_sgen_power=0.005, min_storage_power=0.005,
                           batch_size=1, defer_device=True, seed=0, reference_faithful=True)
    assert lv.init == 'flat' and lv.reference_deviations == {}
