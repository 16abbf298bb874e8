"""Benchmark of the hot path: env.step()/s of the batched, fused step kernel.

    python bench.py [--config C] --gpus N --steps K --warmup W

`--config` selects one of BASELINE.json's configurations (default 2, the one the headline metric is
quoted on):
  1  MaxRenewable,            1-LV-rural1--0-sw (15 buses),  batch 1: the plumbing case (launch-latency bound)
  2  VoltageControl,          1-MV-urban--0-sw (144 buses),  8192 instances per GPU            (weak)
  3  EcoDispatch,             1-HV-mixed--0-sw (306 buses),  8192 instances per GPU            (weak)
  4  QMarket,                 1-MV-urban--0-sw (144 buses),  65536 instances sharded over N    (strong)
  5  N-1 VoltageControl,      1-HV-urban--0-sw (372 buses),  4096 instances x (1 + K) solves,
                              K = every non-islanding line, sharded over N                     (strong)
All grids are synthetic stand-ins of the named SimBench codes (SimBench is unavailable offline).

One "step" = one `env.step()` of the whole batch = ONE launch of the fused kernel (apply actions -> NR
power flow [-> K contingency solves] -> results -> objective -> violations -> reward -> observation).
Inputs are resident in HBM when the timed region starts.

`--gpus N` with N > 1: one process per GPU over torch.distributed (nccl = RCCL).  Launched by
`torch.distributed.run` the ranks are taken from the environment; launched plainly, this script starts
the N ranks itself (child processes, before anything touches a GPU) and relays rank 0's line.  Instances
are independent, so the data path has NO collective; for N > 1 the rewards are all-gathered per step
(`--gather reward`, the default; `obs` adds the observations, `none` removes it) — the one exchange a
centralised learner needs.  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]

import numpy as np  # noqa: E402

# config -> (env class, constructor kwargs, batch, scaling, golden scenario of the CPU oracle)
CONFIGS = {
    1: ('MaxRenewable', dict(simbench_network_name='1-LV-rural1--0-sw', min_sgen_power=0.005, min_storage_power=0.005),
        1, 'weak', 'maxren_lv'),
    2: ('VoltageControl', dict(simbench_network_name='1-MV-urban--0-sw'), 8192, 'weak', 'vc_mv_urban'),
    3: ('EcoDispatch', dict(simbench_network_name='1-HV-mixed--0-sw'), 8192, 'weak', 'eco_hv_mixed'),
    4: ('QMarket', dict(simbench_network_name='1-MV-urban--0-sw'), 65536, 'strong', 'qm_mv_urban'),
    5: ('SecurityConstrainedVoltageControl', dict(simbench_network_name='1-HV-urban--0-sw', n_minus_one_lines='all'),
        4096, 'strong', 'sc_vc_hv_urban'),
}


def byte_model(env, mean_it_per_solve, solves_per_step):
    """ALGORITHMIC bytes per instance-step:
    B_step = 8*(n_in + n_out) + it*8*(2*nnzJ + 2*nnzLU + 4*nJ + 4*nb), `it` = NR iterations of ALL solves
    of the step (base case + contingencies)."""
    info = env.plan.info
    net = env.net
    n_price = sum(1 for k in env.store.dynamic if k[0] in ('poly_cost', 'pwl_cost'))
    n_in = env.n_actions + 2 * len(net.load) + len(net.sgen) + len(net.gen) + len(net.storage) + n_price
    n_out = 2 * info['nb'] + 2 * info['nbr'] + env.n_obs_raw + 3 * env.n_constraints + 4
    n_j = info['npv'] + 2 * info['npq']
    nnz_lu = 4 * info['n_blk']
    per_it = 8 * (2 * info['nnz_j'] + 2 * nnz_lu + 4 * n_j + 4 * info['nb'])
    it_step = mean_it_per_solve * solves_per_step
    b_step = 8 * (n_in + n_out) + it_step * per_it
    # FP64 operation count of the same step (for the "what actually bounds it" report):
    # per NR iteration  A: 24 per off-diagonal Ybus entry + 40 per bus,  B: 46 per update term,
    # C: 8 per U-term + 20 per pivot,  D: 40 per bus;  one more phase A per solve for the final check.
    nnz_off = info['nnz_y'] - info['nb']
    n_piv = info['nb'] - info['nref']
    a_flops = 24 * nnz_off + 40 * info['nb']
    flops_it = a_flops + 46 * info['n_sources'] + 8 * info['n_uterms'] + 20 * n_piv + 40 * info['nb']
    flops = it_step * flops_it + solves_per_step * (a_flops + 10 * info['nbr'] * 4 + 30 * info['nb'])
    return dict(nb=info['nb'], nbr=info['nbr'], nJ=n_j, nnzJ=info['nnz_j'], nnzLU=nnz_lu, levels=info['n_levels'],
                n_in=n_in, n_out=n_out, it_per_solve=mean_it_per_solve, solves_per_step=solves_per_step,
                it=it_step, io_bytes=8 * (n_in + n_out), bytes_per_iteration=per_it, B_step=b_step,
                fp64_flops_per_step=flops)


# ---------------------------------------------------------------------------------------------------
# CPU baseline: the oracle (numpy/SciPy restatement of the reference path: pandas tables + SuperLU
# Newton) on this box's host cores, in worker processes (one, then one per host core up to 32)
# ---------------------------------------------------------------------------------------------------
def cpu_worker(scenario, budget_s, seed):
    """One core: env.step() of the oracle environment for `budget_s` seconds; prints one JSON line."""
    from env_cases import oracle_env, product_env
    host = product_env(scenario, defer_device=True)
    orc = oracle_env(scenario, host)
    n_uniform = host.ops.n_uniform          # price draws of the `_sampling` tails
    rng = np.random.default_rng(seed)
    pool = np.arange(2000, 30000)
    n_act = sum(len(i) for _, _, i in orc.act_keys)
    t_step, n = 0.0, 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget_s or n == 0:
        orc.reset(int(rng.choice(pool)), rng.random(n_uniform))
        a = rng.random(n_act)
        t1 = time.perf_counter()
        out = orc.step(a)
        t_step += time.perf_counter() - t1
        n += bool(out['converged'])
    print(json.dumps({'steps': n, 'cpu_s': t_step}))


def cpu_check_worker(path):
    """Checker leg: the oracle's base-case power flow for the instances described in `path` (steps, draws, actions
    of the first instances of one GPU evaluation); prints their bus voltage magnitudes."""
    from env_cases import oracle_env, product_env
    from oracle import env_oracle
    d = json.load(open(path))
    host = product_env(d['scenario'], defer_device=True)
    orc = oracle_env(d['scenario'], host)
    out = []
    for k in range(len(d['steps'])):
        orc.reset(int(d['steps'][k]), np.asarray(d['uniform'][k], float) if d['uniform'] is not None else ())
        env_oracle.apply_actions(orc.net, orc.act_keys, np.asarray(d['actions'][k], float), orc.autoscale, orc.diff_step)
        ok = orc.solve()
        vm = orc.net['res_bus']['vm_pu'].to_numpy(float) if ok else np.full(len(orc.net['bus']), np.nan)
        out.append([None if v != v else float(v) for v in vm])
    print(json.dumps({'vm': out}))


def voltage_check(env, config, n_check):
    """max |V| error of the GPU path against the CPU oracle on `n_check` instances of one fresh evaluation (the second
    half of BASELINE.json's metric; pandapower itself is not installed, so the oracle stands in — as the checker)."""
    import tempfile
    rng = np.random.default_rng(99)
    B = env.B
    steps = rng.choice(env.train_steps, B)
    uniform = rng.random((B, env.n_uniform)) if env.n_uniform else None
    actions = rng.random((B, env.n_actions))
    env.reset(options={'step': steps, 'uniform': uniform})
    env.step(actions)
    vm_gpu = env.result_table('bus', 'vm_pu')[:n_check].cpu().numpy()          # (result bank: the base case)
    with tempfile.NamedTemporaryFile('w', suffix='.json', delete=False) as fh:
        json.dump({'scenario': CONFIGS[config][4], 'steps': steps[:n_check].tolist(),
                   'uniform': uniform[:n_check].tolist() if uniform is not None else None,
                   'actions': actions[:n_check].tolist()}, fh)
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-check', fh.name], stdout=subprocess.PIPE,
                           stderr=subprocess.DEVNULL, text=True)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
        if not lines:
            return None
        vm_cpu = np.array([[np.nan if v is None else v for v in row] for row in json.loads(lines[-1])['vm']])
    finally:
        os.unlink(fh.name)
    both = np.isfinite(vm_cpu) & np.isfinite(vm_gpu)
    if not both.any() or (np.isfinite(vm_cpu) != np.isfinite(vm_gpu)).any():
        return None
    return float(np.abs(vm_cpu - vm_gpu)[both].max())


def cpu_baseline(config, budget_s=float(os.environ.get('OPFX_BENCH_CPU_BUDGET', 10.0))):     # (env var: shorter sample in tests)
    scenario = CONFIGS[config][4]
    cores = os.cpu_count() or 1

    def run(n_proc):
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), '--cpu-worker', scenario,
                                   '--cpu-budget', str(budget_s), '--cpu-seed', str(k)],
                                  stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True,
                                  env=dict(os.environ, OMP_NUM_THREADS='1', MKL_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1'))
                 for k in range(n_proc)]
        res = []
        for p in procs:
            out, _ = p.communicate()
            lines = [ln for ln in out.splitlines() if ln.startswith('{')]
            if lines:
                res.append(json.loads(lines[-1]))
        return res
    one = run(1)
    rate1 = one[0]['steps'] / one[0]['cpu_s'] if one else float('nan')
    n_many = min(cores, 32)          # (every worker is a Python process with pandas + SciPy: bounded memory)
    many = run(n_many) if n_many > 1 else one
    rate_p = sum(r['steps'] / r['cpu_s'] for r in many) if many else float('nan')
    return dict(value=rate1, unit='env.step()/s', cores=1, kind='port',
                all_cores=dict(value=rate_p, cores=len(many), host_cores=cores),
                sample=f'{one[0]["steps"] if one else 0} instance-steps of {CONFIGS[config][0]} (scenario {scenario}) with '
                       f'the numpy+SciPy oracle (oracle/env_oracle.py + pf_oracle.py), step() only, '
                       f'{one[0]["cpu_s"] if one else 0:.1f} s of CPU work on one core; then {len(many)} independent '
                       f'processes, one per host core; pandapower itself is not installed')


# ---------------------------------------------------------------------------------------------------
def launch_ranks(args):
    """Plain `python bench.py --gpus N` (N > 1): start the N ranks as children, relay rank 0's line."""
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=None, help='instances per GPU (default: the configuration\'s)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--gather', choices=('none', 'reward', 'obs'), default=None,
                    help='per-step RCCL all-gather of the rewards (and observations) on every rank; '
                         'default: reward when N > 1')
    ap.add_argument('--cpu-worker', default=None, help=argparse.SUPPRESS)
    ap.add_argument('--cpu-budget', type=float, default=10.0, help=argparse.SUPPRESS)
    ap.add_argument('--cpu-seed', type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument('--cpu-check', default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_worker:
        return cpu_worker(args.cpu_worker, args.cpu_budget, args.cpu_seed)
    if args.cpu_check:
        return cpu_check_worker(args.cpu_check)
    if args.gpus > 1 and 'RANK' not in os.environ:
        sys.exit(launch_ranks(args))           # nothing has touched a GPU in this process

    import torch
    import torch.distributed as dist
    from opfgym_amd import capi, dist as odist, envs
    rank, world, local_rank = odist.init_from_env()
    if os.environ.get('OPFX_BENCH_SHARE_GPU'):       # debugging aid: all ranks on GPU 0 (use with gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = f'cuda:{local_rank}'
    cls_name, kw, batch_cfg, scaling, _ = CONFIGS[args.config]
    gather_mode = args.gather or ('reward' if world > 1 else 'none')
    if args.batch is not None:
        B = args.batch
    elif scaling == 'weak':
        B = batch_cfg
    else:                                          # strong: the configuration's total, whole instances per rank
        lo, hi = odist.shard_bounds(batch_cfg, rank, world)
        B = hi - lo
    total_B = B * world if (args.batch is not None or scaling == 'weak') else batch_cfg
    env = getattr(envs, cls_name)(batch_size=B, device=device, seed=rank, **kw)
    rng = np.random.default_rng(1234 + rank)
    env.reset(options={'step': rng.choice(env.train_steps, B)})
    act_rng = np.random.default_rng(4321 + rank)
    actions = torch.as_tensor(act_rng.random((B, env.n_actions)), device=device)
    gather = {'none': (), 'reward': ('reward',), 'obs': ('reward', 'obs')}[gather_mode]

    # the gathers run behind the next step's kernel (opfgym_amd.dist.OverlappedGather): the full batch of step k is
    # available while step k+1 is simulated, as a learner consumes it; the last one is collected after the loop
    g_reward, g_obs = odist.OverlappedGather(world), odist.OverlappedGather(world)

    def one_step():
        obs, reward, term, trunc, info = env.step(actions)
        if world > 1 and gather:
            g_reward.submit(reward)
            if 'obs' in gather:
                g_obs.submit(obs)
        return info

    for _ in range(args.warmup):
        info = one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        info = one_step()
    if world > 1 and gather:                   # the last step's gather belongs to the timed region
        g_reward.flush()
        if 'obs' in gather:
            g_obs.flush()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64,
                          device=device if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    conv = float(info['converged'].double().mean().item())
    solves_per_step = 1 + len(env.contingencies)
    mean_it_total = float(info['total_iterations'].double().mean().item())
    mean_it_base = float(info['iterations'].double().mean().item())
    min_pivot = float(info['min_pivot'].min().item())

    # kernel duration: HIP events on the launch stream around back-to-back launches
    io = env._io(actions, False)
    ms = capi.C.c_float()
    reps = max(5, min(args.steps, 50))
    with torch.cuda.device(device):
        capi.check(capi.lib().opfx_time_steps(env._env_handle, B, capi.C.byref(io),
                                              capi.C.byref(env.solve_opts), reps,
                                              capi._stream(), capi.C.byref(ms)), 'opfx_time_steps')
    kernel_ms = ms.value / reps

    # the cycle of a single-step benchmark environment: reset (device-side sampling) + step
    n_cyc = max(3, min(args.steps, 20))
    for _ in range(2):                         # (first use of the device-side random draws)
        env.reset()
        env.step(actions)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(n_cyc):
        env.reset()
        env.step(actions)
    torch.cuda.synchronize()
    cycle_ms = (time.perf_counter() - t1) / n_cyc * 1e3

    if rank == 0:
        team, lds, per_cu = capi.C.c_int32(), capi.C.c_int64(), capi.C.c_int32()
        capi.check(capi.lib().opfx_env_get_info(env._env_handle, capi.C.byref(team), capi.C.byref(lds), capi.C.byref(per_cu)))
        bm = byte_model(env, mean_it_total / solves_per_step, solves_per_step)
        achieved = bm['B_step'] * B / (kernel_ms * 1e-3) / 1e9
        # HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate
        # runs of this same script, see profiles/README.md) — not measurable in-process
        traffic = traffic_raw = None
        pmc_file = os.path.join(ROOT, 'profiles', 'pmc_latest.json')
        if os.path.exists(pmc_file):
            pmc = json.load(open(pmc_file))
            ent = pmc.get(f'config{args.config}', pmc if args.config == 2 and 'hbm_bytes_per_launch_fetch_x2' in pmc else None)
            if ent and ent.get('batch', 8192) == B:
                traffic = ent['hbm_bytes_per_launch_fetch_x2']
                traffic_raw = ent['hbm_bytes_per_launch_raw']
        kernel_name = f'k_step<{"2" if env.plan.info["n_full"] < env.plan.info["n_blk"] else "1"}|1,{team.value}>'
        # what the launch really has to read and write: the instance rows of the caller's buffers
        buffer_io = {'read': int(B * 8 * (env.nx + env.n_actions)),
                     'write': int(B * (sum(v[0].numel() * v.element_size() for v in env.buf.values()) + 8 * env.n_actions))}
        out = {
            'metric': 'env.step()/s (batched NR power-flow solves/s) at batch 8192',
            'value': total_B * args.steps / elapsed,
            'unit': 'env.step()/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True, 'scaling': scaling if args.batch is None else 'weak', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': f'BASELINE config {args.config}: {cls_name} env, synthetic stand-in for SimBench '
                                   f'{kw["simbench_network_name"]} ({bm["nb"]} buses), batch={total_B} '
                                   f'({B} per GPU), {solves_per_step} NR solve(s) per step, step() only',
                       'baseline_config': args.config, 'batch_per_gpu': B, 'batch_total': total_B,
                       'parallelism': f'shard{world}',
                       'collective': {'none': 'none', 'reward': 'all_gather(reward), overlapped with the next step', 'obs': 'all_gather(reward+obs), overlapped with the next step'}[gather_mode] if world > 1 else 'none',
                       'converged_fraction': conv, 'mean_nr_iterations': mean_it_base,
                       'mean_nr_iterations_all_solves': mean_it_total, 'solves_per_step': solves_per_step,
                       'nr_solves_per_s': total_B * solves_per_step * args.steps / elapsed,
                       'reset_plus_step_ms': cycle_ms, 'episodes_per_s_reset_plus_step': B * world / (cycle_ms * 1e-3),
                       'min_relative_pivot': min_pivot,
                       'tolerance_pu': env.solve_opts.tol, 'byte_model': bm,
                       'kernel_launch': {'waves_per_instance': team.value, 'lds_bytes_per_instance': lds.value,
                                         'instances_per_cu': per_cu.value}},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': 8000.0, 'unit': 'GB/s',
                         'frac': achieved / 8000.0, 'traffic': traffic, 'traffic_unit': 'bytes per launch',
                         'traffic_raw_counters': traffic_raw,
                         'algorithmic_bytes_per_launch': bm['B_step'] * B,
                         'compulsory_io_bytes_per_launch': bm['io_bytes'] * B,
                         'buffer_io_bytes_per_launch': buffer_io,
                         'kernel': kernel_name, 'kernel_ms': kernel_ms,
                         'fp64_tflops_achieved': bm['fp64_flops_per_step'] * B / (kernel_ms * 1e-3) / 1e12,
                         'fp64_vector_peak_tflops': 78.6,
                         'note': 'achieved = SURVEY §8d algorithmic bytes (state streamed through memory once per NR '
                                 'phase) x instances per launch / kernel time; the kernel keeps that state in LDS, so '
                                 'real HBM traffic is ~ io_bytes per instance (`traffic`); what bounds the kernel is a '
                                 'wave\'s own issue / LDS-return rate (profiles/*_sq_counters.txt, DESIGN.md)'},
        }
        # timed after the GPU work, in child processes (one, then one per host core)
        out['cpu_baseline'] = cpu_baseline(args.config) if (world == 1 and not args.no_cpu_baseline) else None
        if out['cpu_baseline'] is not None:
            n_check = 2 if args.config == 5 else 4
            err = voltage_check(env, args.config, min(n_check, B))
            out['config']['max_abs_v_err_pu'] = err
            out['config']['max_abs_v_err_against'] = (f'CPU oracle (oracle/pf_oracle.py), base-case |V| of {min(n_check, B)} instances; '
                                                      'pandapower is not installed here')
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
