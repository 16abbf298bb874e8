"""The bench contract on the GPU box: `bench.py` prints ONE JSON line with the driver's keys, the roofline object and
(unless switched off) the CPU baseline with the |V| check against the oracle."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {'metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
        'dtype', 'data', 'config', 'roofline', 'cpu_baseline'}


def _run(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + list(args), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=900, env=dict(os.environ, OPFX_BENCH_CPU_BUDGET='1'))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                       # exactly one line on stdout
    return json.loads(lines[0])


@pytest.mark.parametrize('config', [2])
def test_bench_line_has_the_contract_keys(config):
    d = _run('--config', str(config), '--steps', '4', '--warmup', '2', '--no-cpu-baseline')
    assert set(d) == KEYS
    assert d['n_gpus'] == 1 and d['steps'] == 4 and d['warmup'] == 2 and d['higher_is_better'] is True
    assert d['dtype'] == 'f64' and d['data'] == 'synthetic' and d['vs_baseline'] is None and d['cpu_baseline'] is None
    assert d['value'] > 0 and abs(d['value'] - d['config']['batch_total'] / (d['ms_per_step'] * 1e-3)) < 1e-6 * d['value']
    assert d['config']['baseline_config'] == config and 'workload' in d['config'] and d['config']['converged_fraction'] > 0.99
    r = d['roofline']
    # the bound is the resource with the largest fraction of its own peak, and no fraction of a real resource exceeds 1
    res = r['resources']
    assert r['bound'] in res and r['bound'] == max(res, key=lambda k: res[k]['frac'])
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12 and 0 < r['frac'] <= 1.0
    for name, v in res.items():
        assert abs(v['frac'] - v['achieved'] / v['peak']) < 1e-12 and 0 <= v['frac'] <= 1.0, name
    assert {'hbm', 'valu_issue', 'lds_array', 'lds_bytes', 'fp64_vector'} <= set(res)       # (config 2 has committed counters)
    assert r['hbm_measured_frac'] == res['hbm']['frac'] and r['traffic'] > 0
    # SURVEY §8d's figure, unchanged arithmetic, kept apart from the bound
    ae = r['algorithmic_equiv']
    assert ae['peak'] == 8000.0 and ae['unit'] == 'GB/s'
    assert abs(ae['achieved'] - ae['algorithmic_bytes_per_launch'] / (r['kernel_ms'] * 1e-3) / 1e9) < 1e-6 * ae['achieved']
    # the kernel time comes from the timed region itself: never longer than the wall time per step
    assert r['kernel_ms'] <= d['ms_per_step'] * 1.02
    lb = res['lds_bytes']
    assert abs(lb['wave_instructions_per_launch_model'] / lb['wave_instructions_per_launch_measured'] - 1) < 0.15
    bm = d['config']['byte_model']
    assert abs(bm['B_step'] - (bm['io_bytes'] + bm['it'] * bm['bytes_per_iteration'])) < 1e-6


def test_bench_default_line_carries_cpu_baseline_and_voltage_check():
    d = _run('--config', '1', '--steps', '4', '--warmup', '2')
    cb = d['cpu_baseline']
    assert cb['kind'] == 'port' and cb['cores'] == 1 and cb['value'] > 0 and cb['all_cores']['cores'] >= 1
    assert 'oracle' in cb['sample']
    assert d['config']['max_abs_v_err_pu'] is not None and d['config']['max_abs_v_err_pu'] < 1e-9


def test_two_ranks_on_the_one_gpu_run_the_multi_gpu_path_end_to_end(tmp_path):
    """The path the driver calls on an 8-GPU node — self-launch of the ranks, strong sharding of a BASELINE batch,
    the overlapped all-gather of the rewards — run once on hardware: two ranks sharing GPU 0 (OPFX_BENCH_SHARE_GPU,
    gloo staging through the host).  Rank 0's gathered rewards must be the concatenation of what each shard computes
    on its own in a single-process run with that rank's seeds.  No scaling number is read off this."""
    import numpy as np
    env = dict(os.environ, OPFX_BENCH_SHARE_GPU='1', OPFX_DIST_BACKEND='gloo', OPFX_BENCH_CPU_BUDGET='1')
    full = tmp_path / 'full.npy'
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--config', '4', '--batch', '4096',
                        '--steps', '3', '--warmup', '1', '--dump-reward', str(full)], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['config']['batch_total'] == 8192 and d['config']['batch_per_gpu'] == 4096
    assert d['config']['parallelism'] == 'shard2' and 'all_gather(reward)' in d['config']['collective']
    assert d['cpu_baseline'] is None and d['value'] > 0
    got = np.load(full)
    assert got.shape == (8192,) and np.isfinite(got).all()
    for rank in (0, 1):
        part = tmp_path / f'part{rank}.npy'
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--config', '4', '--batch', '4096',
                            '--steps', '3', '--warmup', '1', '--no-cpu-baseline', '--as-rank', str(rank), '--dump-reward', str(part)],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=os.environ.copy())
        assert r.returncode == 0, r.stderr[-3000:]
        assert np.array_equal(np.load(part), got[rank * 4096:(rank + 1) * 4096]), rank
