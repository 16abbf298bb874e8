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
        'dtype', 'data', 'config', 'roofline', 'cpu_baseline', 'timing', 'dist_backend', 'world_seen', 'gather_check'}


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
    assert d['dist_backend'] is None and d['world_seen'] == 1 and d['gather_check'] is None      # (a plain single process)
    assert d['n_gpus'] == 1 and d['steps'] == 4 and d['warmup'] == 2 and d['higher_is_better'] is True
    assert d['dtype'] == 'f64' and d['data'] == 'synthetic' and d['vs_baseline'] is None and d['cpu_baseline'] is None
    assert d['value'] > 0 and abs(d['value'] - d['config']['batch_total'] / (d['ms_per_step'] * 1e-3)) < 1e-6 * d['value']
    assert d['config']['baseline_config'] == config and 'workload' in d['config'] and d['config']['converged_fraction'] > 0.99
    r = d['roofline']
    # the bound is the resource with the largest fraction of its own peak, and no fraction of a real resource exceeds 1
    res = r['resources']
    real = {k: v for k, v in res.items() if k != 'fp64_useful_of_valu_busy'}      # (that one is a ratio of two of the others)
    assert r['bound'] in real and r['bound'] == max(real, key=lambda k: real[k]['frac'])
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12 and 0 < r['frac'] <= 1.0
    for name, v in res.items():
        assert abs(v['frac'] - v['achieved'] / v['peak']) < 1e-12 and 0 <= v['frac'] <= 1.0, name
    assert {'lds_bytes', 'fp64_vector'} <= set(res)
    if not r['counters_dropped_as_stale']:     # (config 2 has committed counters, taken from sources with this source_sha16)
        assert {'hbm', 'valu_issue', 'lds_array', 'fp64_useful_of_valu_busy'} <= set(res)
        assert r['hbm_measured_frac'] == res['hbm']['frac'] and r['traffic'] > 0 and r['counters_source_sha16'] == r['source_sha16']
        assert abs(res['fp64_useful_of_valu_busy']['frac'] - res['fp64_vector']['frac'] / res['valu_issue']['frac']) < 1e-12
        lb = res['lds_bytes']
        assert abs(lb['wave_instructions_per_launch_model'] / lb['wave_instructions_per_launch_measured'] - 1) < 0.15
    # the headline is the median of five windows of --steps launches each
    t = d['timing']
    assert t['windows'] == 5 and t['steps_per_window'] == 4 and len(t['ms_per_step_windows']) == 5
    assert t['ms_per_step_p10'] <= d['ms_per_step'] <= t['ms_per_step_p90'] and sorted(t['ms_per_step_windows'])[2] == d['ms_per_step']
    # SURVEY §8d's figure, unchanged arithmetic, kept apart from the bound
    ae = r['algorithmic_equiv']
    assert ae['peak'] == 8000.0 and ae['unit'] == 'GB/s'
    assert abs(ae['achieved'] - ae['algorithmic_bytes_per_launch'] / (r['kernel_ms'] * 1e-3) / 1e9) < 1e-6 * ae['achieved']
    # the kernel time comes from the timed region itself: never longer than the wall time per step
    assert r['kernel_ms'] <= d['ms_per_step'] * 1.02
    bm = d['config']['byte_model']
    assert abs(bm['B_step'] - (bm['io_bytes'] + bm['it'] * bm['bytes_per_iteration'])) < 1e-6


def test_bench_default_line_carries_cpu_baseline_voltage_check_and_the_other_configs():
    """The default line (here with config 1 as the headline, so that the block holds 2-5): the CPU baseline, the |V| check
    against the oracle, and `also` — every other BASELINE configuration measured briefly on the same GPU, each with its
    rate, kernel time, convergence, iterations, bound and its own |V| check over 64 (N-1: 8) instances."""
    d = _run('--config', '1', '--steps', '4', '--warmup', '2')
    assert set(d) == KEYS | {'also', 'reference_settings'}
    cb = d['cpu_baseline']
    assert cb['kind'] == 'port' and cb['cores'] == 1 and cb['value'] > 0 and cb['all_cores']['cores'] >= 1
    assert 'oracle' in cb['sample']
    assert d['config']['max_abs_v_err_pu'] is not None and d['config']['max_abs_v_err_pu'] < 1e-9
    assert set(d['also']) == {'config2', 'config3', 'config4', 'config5'}
    for name, a in d['also'].items():
        assert 'error' not in a, (name, a)
        assert a['value'] > 0 and a['kernel_ms'] <= a['ms_per_step'] * 1.02 and a['converged_fraction'] > 0.99, name
        assert 2.0 < a['mean_nr_iterations'] < 8.0 and a['roofline']['bound'] in a['roofline']['fractions'], name
        assert a['max_abs_v_err_pu'] is not None and a['max_abs_v_err_pu'] < 1e-9, name
        assert a['max_abs_v_err_instances'] == (8 if name == 'config5' else 64)
    assert d['also']['config5']['steps'] == 3 and d['also']['config3']['steps'] >= 5
    assert all(a['newton_start'] == 'flat' and a['contingency_start'] == 'base_case' for a in d['also'].values())
    # the plain kernels are specialised on what each environment fixes for the whole batch (no PV bus = 1, no modifiers = 2)
    assert [d['also'][f'config{c}']['kernel'] for c in (2, 3, 4, 5)] == ['k_step<2,1,SPEC=3>', 'k_step<2,4,SPEC=2,MINW=3>', 'k_step<2,1,SPEC=3>', 'k_step<2,4,SPEC=1>']
    # (config 3: three teams of four per CU on the plan with shared LDS slots, BatchedOpfEnv._try_shared_slots)
    assert d['also']['config3']['kernel_launch']['instances_per_cu'] == 3 and d['also']['config5']['kernel_launch']['instances_per_cu'] == 2
    assert d['also']['config3']['shared_slots'] > 150 and d['also']['config5']['shared_slots'] == 0
    # the same workloads at the reference's own solver settings: pandapower's init='auto' is 'dc' on all of them (every
    # stand-in hangs on 110 kV or above), contingencies from scratch; same fixed point -> same |V| check
    rs = d['reference_settings']
    assert set(rs) == {'what', 'config2', 'config3', 'config5'}
    for name in ('config2', 'config3', 'config5'):
        a = rs[name]
        assert 'error' not in a, (name, a)
        assert a['newton_start'] == 'dc' and a['contingency_start'] == 'flat' and ',DC,' in a['kernel'], (name, a['kernel'])
        assert a['value'] > 0 and a['converged_fraction'] > 0.99 and a['max_abs_v_err_pu'] is not None and a['max_abs_v_err_pu'] < 1e-9, name
        assert a['roofline']['counters_from'] is None          # (the committed counters are the plain kernel's)
    # (the DC-start kernels are specialised and compiled for three wavefronts per SIMD like the plain ones)
    assert [rs[f'config{c}']['kernel'] for c in (2, 3, 5)] == ['k_step<2,1,DC,SPEC=3>', 'k_step<2,4,DC,SPEC=2,MINW=3>', 'k_step<2,4,DC,SPEC=1>']
    # contingencies from scratch cost iterations: more of them than from the base case's solution
    assert rs['config5']['mean_nr_iterations_all_solves'] > d['also']['config5']['mean_nr_iterations_all_solves']


def test_the_rccl_path_runs_on_hardware_with_a_world_of_one(tmp_path):
    """RCCL itself, executed: one rank under torchrun with OPFX_FORCE_COLLECTIVE=1 does not short-circuit —
    `init_process_group('nccl', device_id=...)`, the barriers and the max-reduce of the timed windows, the asynchronous
    `all_gather_into_tensor` of rewards and observations on device tensors (OverlappedGather) and `destroy_process_group`
    all run on the nccl (= RCCL) backend.  The gathered rewards must be the local ones."""
    import numpy as np
    env = dict(os.environ, OPFX_FORCE_COLLECTIVE='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('OPFX_DIST_BACKEND', None)
    full, part = tmp_path / 'full.npy', tmp_path / 'part.npy'
    args = ['--gpus', '1', '--config', '2', '--batch', '512', '--steps', '3', '--warmup', '1', '--windows', '2', '--no-cpu-baseline']
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=1', '--master-addr', '127.0.0.1',
                        '--master-port', '29533', os.path.join(ROOT, 'bench.py')] + args + ['--gather', 'obs', '--dump-reward', str(full)],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['dist_backend'] == 'nccl' and d['world_seen'] == 1 and d['n_gpus'] == 1
    assert d['config']['collective'] == 'all_gather(reward+obs), overlapped with the next step'
    assert d['gather_check'] == {'rows': 512, 'expected_rows': 512, 'local_shard_identical': True}
    got = np.load(full)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args + ['--dump-reward', str(part)],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=os.environ.copy())
    assert r.returncode == 0, r.stderr[-3000:]
    assert got.shape == (512,) and np.isfinite(got).all() and np.array_equal(np.load(part), got)


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


def _ranks_on_one_gpu(tmp_path, n_ranks, args, compare_ranks, shard_of):
    """`bench.py --gpus n_ranks <args>` with every rank on GPU 0 (gloo staging through the host); rank 0's gathered rewards
    against single-process runs of the ranks in `compare_ranks` with that rank's seeds and shard."""
    import numpy as np
    env = dict(os.environ, OPFX_BENCH_SHARE_GPU='1', OPFX_DIST_BACKEND='gloo', OPFX_BENCH_CPU_BUDGET='1')
    full = tmp_path / 'full.npy'
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n_ranks)] + args +
                       ['--steps', '2', '--warmup', '1', '--windows', '1', '--dump-reward', str(full)], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=1500, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    got = np.load(full)
    for rank in compare_ranks:
        part = tmp_path / f'part{rank}.npy'
        lo, hi, extra = shard_of(rank)
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1'] + args + extra +
                           ['--steps', '2', '--warmup', '1', '--windows', '1', '--no-cpu-baseline', '--as-rank', str(rank),
                            '--dump-reward', str(part)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900,
                           env=os.environ.copy())
        assert r.returncode == 0, r.stderr[-3000:]
        assert np.array_equal(np.load(part), got[lo:hi]), rank
    return d, got


def test_world_of_eight_on_the_one_gpu_weak(tmp_path):
    """The command lines the driver issues on the 8-GPU node, end to end at world size 8 once: weak
    scaling, `--gpus 8 --config 2 --batch 64` and `--gpus 8 --config 4 --batch 512`.  No scaling number is read off this."""
    import numpy as np
    for cfg, b in ((2, 64), (4, 512)):
        sub = tmp_path / f'c{cfg}'
        sub.mkdir()
        d, got = _ranks_on_one_gpu(sub, 8, ['--config', str(cfg), '--batch', str(b)], (0, 7),
                                   lambda rank, b=b: (rank * b, (rank + 1) * b, []))
        assert d['n_gpus'] == 8 and d['scaling'] == 'weak' and d['config']['batch_per_gpu'] == b and d['config']['batch_total'] == 8 * b
        assert d['config']['parallelism'] == 'shard8' and d['config']['collective'] == 'all_gather(reward), overlapped with the next step'
        assert got.shape == (8 * b,) and np.isfinite(got).all() and d['value'] > 0 and d['cpu_baseline'] is None


def test_world_of_seven_on_the_one_gpu_strong_with_ragged_shards(tmp_path):
    """`--gpus 7 --config 4`: BASELINE config 4's 65 536 instances sharded over a world size that does not divide them —
    shards of 9 363 and 9 362 whole instances, padded for the collective and trimmed on hand-over."""
    import numpy as np
    from opfgym_amd.dist import shard_bounds
    d, got = _ranks_on_one_gpu(tmp_path, 7, ['--config', '4'], (0, 3, 6),
                               lambda rank: (*shard_bounds(65536, rank, 7), ['--of-world', '7']))
    assert d['n_gpus'] == 7 and d['scaling'] == 'strong' and d['config']['batch_total'] == 65536
    assert d['config']['batch_per_gpu'] == shard_bounds(65536, 0, 7)[1] and d['config']['parallelism'] == 'shard7'
    assert 'all_gather(reward)' in d['config']['collective']
    assert got.shape == (65536,) and np.isfinite(got).all()
