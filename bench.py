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


def lds_model(env, mean_it_per_solve, solves_per_step, team):
    """LDS bytes one instance-step moves, counted from the compiled plan (8 B per active lane and access; an atomic
    add counts once).  Per NR iteration: phase A (bus rounds: own V + ELL-width neighbours read, blocks / rhs
    written; overflow entries), phases B + C (twelve reads + four or two atomic adds per live item, riders, second columns), phase D
    (diagonal block + rhs + V read, V written), fill blocks zeroed; one more phase A per solve for the final mismatch.
    Per step: the table row staged in LDS and read back by the prologue, the result bank written and read.
    Cross-check: `instructions` (wave-level LDS instructions) against SQ_INSTS_LDS of profiles/*_sq_counters.txt."""
    plan, info = env.plan, env.plan.info
    nb, KA = info['nb'], info['lp_ell_width']
    two_value = info['n_full'] < info['n_blk'] and env.kernel_info()['packed']
    a_ent = plan.array('LP_A_ENT').astype(np.int64) & 0xFFFFFFFF
    live_blk = int(((a_ent >> 16) != 0xFFFF).sum())
    h_ent = plan.array('LP_H_ENT').astype(np.int64) & 0xFFFFFFFF
    h_live = int(((h_ent & 0xFFFF) != 0xFFFF).sum())
    vals_off = 2 if two_value else 4                      # values of a plain off-diagonal block
    # phase A, per pass: reads (own vr, vi + KA neighbours) per bus, bus type byte ignored
    a_reads = nb * (2 + 2 * KA) + h_live * 4
    a_writes = live_blk * vals_off + h_live * vals_off + 2 * h_live + nb * (4 + 2) + len(plan.array('LP_H_ROW')) * 2
    a_instr = info['lp_rounds_a'] * (2 + 2 * KA + KA * vals_off + 6) + info['lp_rounds_h'] * (4 + vals_off + 2)
    if team == 1:
        items = plan.array('LP_B').astype(np.int64).reshape(-1, 2) & 0xFFFFFFFF
        riders = plan.array('LP_B2').astype(np.int64) & 0xFFFFFFFF
        itc = plan.array('LP_C').astype(np.int64).reshape(-1, 2) & 0xFFFFFFFF
        live_b, live_c = (items[:, 0] & 0xFFFF) != 0xFFFF, (itc[:, 0] & 0xFFFF) != 0xFFFF
        rhs_b = live_b & ((items[:, 0] & 0x8000) != 0)
        n_rider = int(((riders >> 16) != 0xFFFF).sum())
        n_second = int(((plan.array('LP_B3').astype(np.int64) & 0xFFFF) != 0xFFFF).sum())
        n_items, n_rhs = int(live_b.sum() + live_c.sum()), int(rhs_b.sum() + live_c.sum())
        rounds = info['lp_rounds_b'] + info['lp_rounds_c']
    else:
        tm = plan.array('LP_TEAM4' if team == 4 else 'LP_TEAM2').astype(np.int64).reshape(-1, 4) & 0xFFFFFFFF
        live = (tm[:, 0] & 0xFFFF) != 0xFFFF
        n_items, n_rhs, n_rider = int(live.sum()), int((live & ((tm[:, 0] & 0x8000) != 0)).sum()), 0
        n_second = int((live & ((tm[:, 2] & 0xFFFF) != 0xFFFF)).sum())
        rounds = info[f'team_rounds_{team}'] * team
    bc_reads = 12 * n_items + 2 * n_rider + 4 * n_second           # (a second column: its block read, its target added to)
    bc_writes = 4 * (n_items - n_rhs) + 2 * n_rhs + 2 * n_rider + 4 * n_second
    bc_instr = rounds * (12 + 4) + (info['lp_rounds_b'] * (4 + 8) if team == 1 else 0)
    n_free = nb - info['nref']
    d_reads, d_writes = n_free * 8, n_free * 2
    d_instr = info['lp_rounds_a'] * 11
    fill_w = info['n_fill'] * vals_off
    per_it = 8 * (a_reads + a_writes + bc_reads + bc_writes + d_reads + d_writes + fill_w)
    per_pass_a = 8 * (a_reads + a_writes + fill_w)
    nres = env.n_results
    per_step = 8 * (2 * env.nx + 2 * env.n_actions + 2 * env.n_inj + solves_per_step * (2 * nres + 4 * nb + 6 * info['nbr']))
    it_step = mean_it_per_solve * solves_per_step
    total = it_step * per_it + solves_per_step * per_pass_a + per_step
    instr_it = a_instr + bc_instr + d_instr + (info['n_fill'] * vals_off + 63) // 64
    instr = it_step * instr_it + solves_per_step * a_instr + (2 * env.nx + 2 * env.n_inj + solves_per_step * (2 * nres + 4 * nb + 6 * info['nbr'])) / 64.0
    return dict(bytes_per_iteration=per_it, bytes_per_step=total, items_per_iteration=n_items, riders=n_rider, second_columns=n_second,
                rounds_bc=rounds, lds_wave_instructions_per_step=instr)


def measured_counters(config, B):
    """Counter means of the committed rocprofv3 passes for this configuration (profiles/pmc_latest.json, written by
    scripts/profile_round.sh: HBM bytes and SQ counters cannot be read in-process), or {} when the batch differs."""
    pmc_file = os.path.join(ROOT, 'profiles', 'pmc_latest.json')
    if not os.path.exists(pmc_file):
        return {}
    ent = json.load(open(pmc_file)).get(f'config{config}')
    return ent if ent and ent.get('batch', 8192) == B else {}


# ---------------------------------------------------------------------------------------------------
# CPU baseline: the oracle (numpy/SciPy restatement of the reference path: pandas tables + SuperLU
# Newton) on this box's host cores, in worker processes (one, then one per host core)
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
    # one worker per host core (P = os.cpu_count()); every worker is a Python process with pandas + SciPy
    # (~0.3 GB resident): bounded by the memory that is free, and by OPFX_BENCH_CPU_PROCS for tests
    n_many = cores
    try:
        import psutil
        n_many = max(1, min(n_many, int(psutil.virtual_memory().available / 0.6e9)))
    except Exception:
        pass
    n_many = int(os.environ.get('OPFX_BENCH_CPU_PROCS', n_many))
    many = run(n_many) if n_many > 1 else one
    rate_p = sum(r['steps'] / r['cpu_s'] for r in many) if many else float('nan')
    # (a box's `os.cpu_count()` counts hardware threads and may exceed what the container is allowed to use at once:
    #  the aggregate at 32 processes is reported next to it)
    some = run(32) if n_many > 32 else many
    rate_32 = sum(r['steps'] / r['cpu_s'] for r in some) if some else float('nan')
    # (a box whose container may use fewer CPUs at once than os.cpu_count() reports runs SLOWER with one process per reported
    #  core than with 32: the best aggregate observed is named beside the two)
    best = max(((rate_p, len(many)), (rate_32, len(some)), (rate1, 1)), key=lambda rc: (rc[0] if rc[0] == rc[0] else -1.0))
    return dict(value=rate1, unit='env.step()/s', cores=1, kind='port',
                all_cores=dict(value=rate_p, cores=len(many), host_cores=cores),
                at_32_processes=dict(value=rate_32, cores=len(some)),
                best_aggregate=dict(value=best[0], processes=best[1]),
                sample=f'{one[0]["steps"] if one else 0} instance-steps of {CONFIGS[config][0]} (scenario {scenario}) with '
                       f'the numpy+SciPy oracle (oracle/env_oracle.py + pf_oracle.py), step() only, '
                       f'{one[0]["cpu_s"] if one else 0:.1f} s of CPU work on one core; then {len(many)} independent '
                       f'processes on the {cores} host cores ({"one per core" if len(many) == cores else "fewer than cores: bounded by free memory / OPFX_BENCH_CPU_PROCS"}); '
                       f'pandapower itself is not installed')


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


def source_sha16():
    """Hash of the sources the kernels are built from: profiles/pmc_latest.json records it with the counters, and counters
    taken from another state of the sources are not mixed into a line."""
    import hashlib
    h = hashlib.sha256()
    for rel in ('opfgym_amd/csrc/opfx_dev.h', 'opfgym_amd/csrc/opfx.hip', 'opfgym_amd/csrc/plan.cpp', 'opfgym_amd/csrc/plan.h'):      # (kernels, host side, plan compiler)
        h.update(open(os.path.join(ROOT, rel), 'rb').read())
    return h.hexdigest()[:16]


def percentile(v, q):
    return float(np.percentile(np.asarray(v, float), q))


def timed_windows(one_step, steps, windows, world, device, flush=None, collective=None):
    """`windows` timed regions of exactly `steps` steps each, every one bracketed by a barrier + synchronize on both sides,
    the elapsed time of a window = max over ranks; HIP events on the launch stream around the same regions.  Returns
    (wall seconds per window, device milliseconds per window, the last step's info, what `flush` returned last)."""
    import torch
    import torch.distributed as dist
    wall, dev_ms, info, tail = [], [], None, None
    if collective is None:                     # (a forced world of one runs the barrier and the max-reduce for real)
        collective = world > 1
    for _ in range(windows):
        torch.cuda.synchronize()
        if collective:
            dist.barrier()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for _ in range(steps):
            info = one_step()
        ev1.record()
        if flush is not None:                  # (the last step's gather belongs to the timed region)
            tail = flush()
        torch.cuda.synchronize()
        if collective:
            dist.barrier()
        el = time.perf_counter() - t0
        if collective:
            tt = torch.tensor([el], dtype=torch.float64, device=device if dist.get_backend() == 'nccl' else 'cpu')
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        wall.append(el)
        dev_ms.append(ev0.elapsed_time(ev1))
    return wall, dev_ms, info, tail


def roofline_of(env, config, B, kernel_ms, mean_it_total, solves_per_step, device):
    """The resources that could bound the step kernel, each as achieved / its own peak (DESIGN.md §4 Roofline)."""
    import torch
    ki = env.kernel_info()
    team = ki['waves_per_instance']
    bm = byte_model(env, mean_it_total / solves_per_step, solves_per_step)
    lm = lds_model(env, mean_it_total / solves_per_step, solves_per_step, team)
    kernel_s = kernel_ms * 1e-3
    props = torch.cuda.get_device_properties(device)
    n_cu = int(props.multi_processor_count)
    n_simd, n_xcd = 4 * n_cu, 8
    algorithmic = bm['B_step'] * B / kernel_s / 1e9              # GB/s, SURVEY §8d model
    # ---- what the committed rocprofv3 passes of this configuration measured (profiles/pmc_latest.json) ----
    pm = measured_counters(config, B)
    stale = bool(pm) and pm.get('source_sha16') != source_sha16()
    # the instantiation that ran: the DC start and the chord steps are kernels of their own (other register counts,
    # other occupancy) — the committed counters are those of the plain one and are not mixed into their lines
    # (opfx.hip do_step: the DC start wins over chord steps; the configurations of this bench all run LDS-resident)
    dc = env.init == 'dc'
    chord = env.jacobian_reuse_tol > 0.0 and not dc
    other_kernel = bool(pm) and (dc or chord)
    if stale or other_kernel:                  # counters of another state of the sources / of another kernel
        pm = {}
    sq = pm.get('sq', {})
    traffic, traffic_raw = pm.get('hbm_bytes_per_launch_fetch_x2'), pm.get('hbm_bytes_per_launch_raw')
    cycles = sq['GRBM_GUI_ACTIVE'] / n_xcd if 'GRBM_GUI_ACTIVE' in sq else None      # kernel duration in GPU clocks
    fr = {}
    if traffic is not None:
        # HBM: measured bytes of the profiled launch / its own duration (the profile's, so both from one run)
        prof_s = pm.get('kernel_avg_ns', kernel_ms * 1e6) * 1e-9
        fr['hbm'] = dict(achieved=traffic / prof_s / 1e9, peak=8000.0, unit='GB/s')
    if cycles:
        # SQ_ACTIVE_INST_VALU counts in units of 4 cycles (quad-cycles), per SIMD; SQ_LDS_IDX_ACTIVE in LDS-array
        # cycles, per CU (MI355X_MICROARCH.md, LDS section; VERDICT r02 recomputation)
        fr['valu_issue'] = dict(achieved=sq['SQ_ACTIVE_INST_VALU'] * 4 / (n_simd * cycles), peak=1.0,
                                unit='fraction of SIMD cycles with a vector instruction in issue')
        fr['lds_array'] = dict(achieved=sq['SQ_LDS_IDX_ACTIVE'] / (n_cu * cycles), peak=1.0,
                               unit='fraction of LDS-array cycles busy',
                               bank_conflict_share=sq.get('SQ_LDS_BANK_CONFLICT', 0.0) / sq['SQ_LDS_IDX_ACTIVE'])
    for v in fr.values():
        v['frac'] = v['achieved'] / v['peak']
    # LDS bytes from the plan (live) against the guide's aggregate ds_read_b64 rate
    lds_rate = lm['bytes_per_step'] * B / kernel_s / 1e12
    fr['lds_bytes'] = dict(achieved=lds_rate, peak=150.0, unit='TB/s', frac=lds_rate / 150.0,
                           bytes_per_launch=lm['bytes_per_step'] * B,
                           wave_instructions_per_launch_model=lm['lds_wave_instructions_per_step'] * B,
                           wave_instructions_per_launch_measured=sq.get('SQ_INSTS_LDS'))
    fp64 = bm['fp64_flops_per_step'] * B / kernel_s / 1e12
    fr['fp64_vector'] = dict(achieved=fp64, peak=78.6, unit='TFLOP/s', frac=fp64 / 78.6)
    if 'valu_issue' in fr:
        # what share of the cycles the vector ALU is busy is FP64 arithmetic of the algorithm (the rest: addresses,
        # descriptor unpacking, selects, LDS plumbing — DESIGN.md §4 "What the VALU cycles are")
        fr['fp64_useful_of_valu_busy'] = dict(achieved=fr['fp64_vector']['frac'] / fr['valu_issue']['frac'], peak=1.0,
                                              unit='fp64_vector.frac / valu_issue.frac', frac=fr['fp64_vector']['frac'] / fr['valu_issue']['frac'])
    # the binding resource: the largest fraction of its own peak (none of them can exceed 1)
    cand = {k: v['frac'] for k, v in fr.items() if k != 'fp64_useful_of_valu_busy'}
    bound = max(cand, key=cand.get)
    # (template arguments as rocprofv3 prints them: block storage, wavefronts per instance, then DC / CHORD or — the plain
    #  kernels — SPEC=n, the specialisation on what the environment fixes for the whole batch, opfx_env_get_spec)
    # (the full-Newton kernels, plain and with the DC start, are specialised and exist for three wavefronts per SIMD; the
    #  chord kernels are neither)
    kernel_name = f'k_step<{2 if ki["packed"] else 1},{team}' + (',DC' if dc else '') + (',CHORD' if chord else '') + \
        (f',SPEC={ki["spec"]}' if not chord else '') + \
        (',MINW=3' if not chord and ki['waves_per_instance'] * ki['instances_per_cu'] > 8 and not (dc and team == 1) else '') + '>'    # (the instantiation compiled for three wavefronts per SIMD)
    # what the launch really has to read and write: the instance rows of the caller's buffers
    buffer_io = {'read': int(B * 8 * (ki.get('x_columns_read', env.nx) + env.n_actions)),      # (the row up to the last column the step names)
                 'write': int(B * (sum(v[0].numel() * v.element_size() for v in env.buf.values()) + 8 * env.n_actions))}
    roof = {
        'bound': bound, 'achieved': fr[bound]['achieved'], 'peak': fr[bound]['peak'], 'unit': fr[bound]['unit'],
        'frac': fr[bound]['frac'],
        'traffic': traffic, 'traffic_unit': 'bytes per launch', 'traffic_raw_counters': traffic_raw,
        'kernel': kernel_name, 'kernel_ms': kernel_ms,
        'kernel_ms_source': 'HIP events on the launch stream around the timed window (median window) / steps: device time of '
                            'the Python step loop, host gaps between launches included',
        'resources': fr,
        'hbm_measured_frac': fr['hbm']['frac'] if 'hbm' in fr else None,
        # SURVEY §8d's figure, kept with unchanged arithmetic: the bytes a memory-streaming Newton solver would
        # move, divided by this kernel's time.  It is NOT a fraction of anything this kernel saturates (the
        # iteration state lives in LDS) and may exceed 1
        'algorithmic_equiv': {'achieved': algorithmic, 'peak': 8000.0, 'unit': 'GB/s', 'frac': algorithmic / 8000.0,
                              'algorithmic_bytes_per_launch': bm['B_step'] * B,
                              'compulsory_io_bytes_per_launch': bm['io_bytes'] * B},
        'buffer_io_bytes_per_launch': buffer_io,
        'counters_from': pm.get('tag'), 'counters_source_sha16': pm.get('source_sha16'), 'source_sha16': source_sha16(),
        'counters_dropped_as_stale': stale, 'counters_dropped_as_other_kernel': other_kernel,
        'hbm_target_note': 'north_star asks for >= 40 % of the HBM roofline: structurally n/a for this design — the Newton '
                           'state is LDS-resident, HBM carries inputs + outputs only (hbm_measured_frac); the binding '
                           'resource and its fraction are what `bound` / `frac` report',
        'note': 'bound = the resource with the largest fraction of its own peak.  hbm / valu_issue / lds_array come '
                'from the committed rocprofv3 passes of this configuration (profiles/pmc_latest.json <- '
                'scripts/profile_round.sh, taken from sources with the same source_sha16; recompute from '
                'profiles/<tag>_sq_counters.txt: VALU = SQ_ACTIVE_INST_VALU*4 / (SIMDs * GRBM_GUI_ACTIVE/8), LDS = '
                'SQ_LDS_IDX_ACTIVE / (CUs * GRBM_GUI_ACTIVE/8)); lds_bytes, fp64_vector and algorithmic_equiv are computed '
                'live from the plan and this run\'s kernel time'}
    return roof, bm, lm, ki, n_cu, n_simd


def also_config(config, device, steps, warmup, **env_kw):
    """One of the BASELINE configurations, measured briefly on this GPU after the headline: the
    driver's one line then carries all five.  `env_kw`: environment settings on top of the configuration's (the
    `reference_settings` block passes reference_faithful=True)."""
    import torch
    from opfgym_amd import envs
    cls_name, kw, B, scaling, _ = CONFIGS[config]
    t_build = time.perf_counter()
    env = getattr(envs, cls_name)(batch_size=B, device=device, seed=0, **kw, **env_kw)
    rng = np.random.default_rng(1234)
    opts = {'step': rng.choice(env.train_steps, B)}
    if env.n_uniform:
        opts['uniform'] = rng.random((B, env.n_uniform))
    env.reset(options=opts)
    actions = torch.as_tensor(np.random.default_rng(4321).random((B, env.n_actions)), device=device)
    t_build = time.perf_counter() - t_build
    for _ in range(warmup):
        env.step(actions)
    wall, dev_ms, info, _ = timed_windows(lambda: env.step(actions)[4], steps, 1, 1, device)
    solves_per_step = 1 + len(env.contingencies)
    mean_it_total = float(info['total_iterations'].double().mean().item())
    kernel_ms = dev_ms[0] / steps
    roof, bm, lm, ki, _, _ = roofline_of(env, config, B, kernel_ms, mean_it_total, solves_per_step, device)
    n_check = 8 if config == 5 else 64
    err = voltage_check(env, config, min(n_check, B))
    out = {'workload': f'{cls_name}, {kw["simbench_network_name"]} stand-in ({bm["nb"]} buses), batch={B}, '
                       f'{solves_per_step} NR solve(s) per step',
           'value': B * steps / wall[0], 'unit': 'env.step()/s', 'steps': steps, 'warmup': warmup,
           'ms_per_step': wall[0] / steps * 1e3, 'kernel_ms': kernel_ms, 'kernel': roof['kernel'],
           'newton_start': env.init, 'contingency_start': 'flat' if env.solve_opts.contingency_start else 'base_case',
           'kernel_launch': {k: ki[k] for k in ('waves_per_instance', 'lds_bytes_per_instance', 'instances_per_cu')},
           'shared_slots': env.plan.info['n_shared'],
           'nr_solves_per_s': B * solves_per_step * steps / wall[0],
           'converged_fraction': float(info['converged'].double().mean().item()),
           'mean_nr_iterations': float(info['iterations'].double().mean().item()),
           'mean_nr_iterations_all_solves': mean_it_total,
           'roofline': {'bound': roof['bound'], 'frac': roof['frac'], 'counters_from': roof['counters_from'],
                        'algorithmic_equiv_frac': roof['algorithmic_equiv']['frac'],
                        'fractions': {k: v['frac'] for k, v in roof['resources'].items()}},
           'max_abs_v_err_pu': err, 'max_abs_v_err_instances': min(n_check, B), 'env_build_s': t_build}
    env.close()
    del env
    torch.cuda.empty_cache()
    return out


def main():
    from opfgym_amd import capi as _capi
    _capi.set_default_debug(_capi.debug_from_env())      # A/B harnesses steer this script through OPFX_* variables; the driver sets none
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--windows', type=int, default=5, help='timed windows of --steps steps each; value = the median window')
    ap.add_argument('--batch', type=int, default=None, help='instances per GPU (default: the configuration\'s)')
    ap.add_argument('--reuse-tol', type=float, default=0.0, help='opfx_solve_opts.jacobian_reuse_tol: chord steps once the mismatch is '
                    'below this (0 = full Newton, the headline and the reference\'s algorithm)')
    ap.add_argument('--init', choices=('flat', 'dc', 'auto'), default='flat', help="start of the Newton iteration (pandapower's `init`; "
                    "'auto' = pandapower's own default: 'dc' on grids fed above 70 kV); the headline runs the flat start")
    ap.add_argument('--no-cpu-baseline', action='store_true', help='GPU measurement only (also skips the `also` block and the |V| check)')
    ap.add_argument('--no-also', action='store_true', help='do not measure the other BASELINE configurations after the headline')
    ap.add_argument('--gather', choices=('none', 'reward', 'obs'), default=None,
                    help='per-step RCCL all-gather of the rewards (and observations) on every rank; '
                         'default: reward when N > 1')
    ap.add_argument('--dump-reward', default=None, help='rank 0 saves the rewards of the last timed step here (.npy): the '
                    'all-gathered full batch for N > 1 with --gather reward, the local batch for N = 1')
    ap.add_argument('--as-rank', type=int, default=None, help='(N = 1) use the seeds rank R of a multi-GPU run uses')
    ap.add_argument('--of-world', type=int, default=None, help='(N = 1, with --as-rank) take the shard rank R of a world of this size owns')
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
    # (switches of this harness, mapped onto arguments: the package itself reads the torchrun variables only)
    rank, world, local_rank = odist.init_from_env(backend=os.environ.get('OPFX_DIST_BACKEND') or None,
                                                  force_collective=os.environ.get('OPFX_FORCE_COLLECTIVE', '') == '1')
    if os.environ.get('OPFX_BENCH_SHARE_GPU'):       # debugging aid: all ranks on GPU 0 (use with gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = f'cuda:{local_rank}'
    cls_name, kw, batch_cfg, scaling, _ = CONFIGS[args.config]
    # the collectives run when there is more than one rank — or when a single rank is told to execute them anyway
    # (OPFX_FORCE_COLLECTIVE=1 under torchrun: the RCCL path on a one-GPU box, tests/test_gpu_bench.py)
    collective = dist.is_initialized() and (world > 1 or odist.force_collective())
    gather_mode = args.gather or ('reward' if collective else 'none')
    seed_rank = rank if args.as_rank is None else args.as_rank
    # --batch N: N instances per GPU (weak scaling), whatever the configuration; without it the configuration's own batch —
    # per GPU for the weak ones, as a TOTAL sharded over the ranks for the strong ones (whole instances per rank; a world
    # size that does not divide it gives ragged shards).  --as-rank R --of-world W (single process): the shard rank R of a
    # world of W ranks owns, with its seeds — what the multi-GPU tests compare the gathered batch against.
    strong_total = batch_cfg if (args.batch is None and scaling == 'strong') else None
    if strong_total is not None:
        w_, r_ = (args.of_world, seed_rank) if (world == 1 and args.of_world) else (world, rank)
        lo, hi = odist.shard_bounds(strong_total, r_, w_)
        B = hi - lo
        total_B = strong_total if world > 1 else B
    else:
        B = args.batch if args.batch is not None else batch_cfg
        total_B = B * world
    env = getattr(envs, cls_name)(batch_size=B, device=device, seed=seed_rank, init=args.init, jacobian_reuse_tol=args.reuse_tol, **kw)
    rng = np.random.default_rng(1234 + seed_rank)
    reset_options = {'step': rng.choice(env.train_steps, B)}
    if env.n_uniform:                       # (explicit draws: a run is then a function of the rank's seed alone)
        reset_options['uniform'] = rng.random((B, env.n_uniform))
    env.reset(options=reset_options)
    act_rng = np.random.default_rng(4321 + seed_rank)
    actions = torch.as_tensor(act_rng.random((B, env.n_actions)), device=device)
    gather = {'none': (), 'reward': ('reward',), 'obs': ('reward', 'obs')}[gather_mode]

    # the gathers run behind the next step's kernel (opfgym_amd.dist.OverlappedGather): the full batch of step k is
    # available while step k+1 is simulated, as a learner consumes it; the last one is collected after the loop
    # (strong-scaling configs on a world size that does not divide the batch give ragged shards: padded and trimmed)
    sizes = [odist.shard_bounds(strong_total, r, world)[1] - odist.shard_bounds(strong_total, r, world)[0] for r in range(world)] \
        if strong_total is not None else None
    g_reward, g_obs = odist.OverlappedGather(world, sizes), odist.OverlappedGather(world, sizes)

    def one_step():
        obs, reward, term, trunc, info = env.step(actions)
        if collective and gather:
            g_reward.submit(reward)
            if 'obs' in gather:
                g_obs.submit(obs)
        return info

    def flush():
        last = g_reward.flush()
        if 'obs' in gather:
            g_obs.flush()
        return last

    for _ in range(args.warmup):
        info = one_step()
    # `windows` windows of EXACTLY --steps steps, each bracketed by a barrier + synchronize on both sides and reduced to
    # the max over ranks; the reported time is the MEDIAN window (a 5 ms region is otherwise one sample), p10 / p90 beside it
    wall, dev_ms, info, last_reward = timed_windows(one_step, args.steps, max(1, args.windows), world, device,
                                                    flush if (collective and gather) else None, collective)
    gather_check = None
    if collective and gather and last_reward is not None:
        # the all-gathered batch of the last step holds this rank's rewards at this rank's rows
        lo_ = sum(sizes[:rank]) if sizes is not None else rank * B
        gather_check = {'rows': int(last_reward.shape[0]), 'expected_rows': int(total_B),
                        'local_shard_identical': bool(torch.equal(last_reward[lo_:lo_ + B].to(env.buf['reward'].device), env.buf['reward']))}
    order = np.argsort(wall)
    mid = int(order[len(order) // 2])
    elapsed = wall[mid]
    kernel_ms = dev_ms[mid] / args.steps
    if args.dump_reward and rank == 0:
        np.save(args.dump_reward, (last_reward if last_reward is not None else env.buf['reward']).cpu().numpy())
    conv = float(info['converged'].double().mean().item())
    solves_per_step = 1 + len(env.contingencies)
    mean_it_total = float(info['total_iterations'].double().mean().item())
    mean_it_base = float(info['iterations'].double().mean().item())
    min_pivot = float(info['min_pivot'].min().item())

    # cross-check: the C ABI's own helper (hipEvents around `reps` launches with nothing else on the stream)
    io = env._io(actions, False)
    ms = capi.C.c_float()
    reps = max(5, min(args.steps, 50))
    with torch.cuda.device(device):
        for _ in range(2):                     # (the first pass brings the clocks back up after the host-side work above)
            capi.check(capi.lib().opfx_time_steps(env._env_handle, B, capi.C.byref(io),
                                                  capi.C.byref(env.solve_opts), reps,
                                                  capi._stream(), capi.C.byref(ms)), 'opfx_time_steps')
    kernel_ms_helper = ms.value / reps

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
        roof, bm, lm, ki, n_cu, n_simd = roofline_of(env, args.config, B, kernel_ms, mean_it_total, solves_per_step, device)
        roof['kernel_ms_back_to_back_helper'] = kernel_ms_helper
        per_step_ms = [w / args.steps * 1e3 for w in wall]
        out = {
            'metric': 'env.step()/s (batched NR power-flow solves/s) at batch 8192',
            'value': total_B * args.steps / elapsed,
            'unit': 'env.step()/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            # what torch.distributed itself saw (None / 1: no process group — a single process without collectives)
            'dist_backend': dist.get_backend() if dist.is_initialized() else None,
            'world_seen': dist.get_world_size() if dist.is_initialized() else 1,
            'gather_check': gather_check,
            'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'strong' if strong_total is not None else 'weak', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'timing': {'windows': len(wall), 'steps_per_window': args.steps, 'reported': 'median window',
                       'ms_per_step_windows': per_step_ms, 'ms_per_step_p10': percentile(per_step_ms, 10),
                       'ms_per_step_p90': percentile(per_step_ms, 90), 'ms_per_step_min': min(per_step_ms)},
            'config': {'workload': f'BASELINE config {args.config}: {cls_name} env, synthetic stand-in for SimBench '
                                   f'{kw["simbench_network_name"]} ({bm["nb"]} buses), batch={total_B} '
                                   f'({B} per GPU), {solves_per_step} NR solve(s) per step, step() only',
                       'baseline_config': args.config, 'batch_per_gpu': B, 'batch_total': total_B,
                       'parallelism': f'shard{world}', 'newton_variant': 'full Newton (a Jacobian factorisation per iteration)' if not args.reuse_tol else
                                         f'Shamanskii / chord steps below a mismatch of {args.reuse_tol:g} (jacobian_reuse_tol; opt-in, NOT the reference\'s algorithm)',
                       'newton_start': env.init,
                       'collective': {'none': 'none', 'reward': 'all_gather(reward), overlapped with the next step', 'obs': 'all_gather(reward+obs), overlapped with the next step'}[gather_mode] if collective else 'none',
                       'converged_fraction': conv, 'mean_nr_iterations': mean_it_base,
                       'mean_nr_iterations_all_solves': mean_it_total, 'solves_per_step': solves_per_step,
                       'nr_solves_per_s': total_B * solves_per_step * args.steps / elapsed,
                       'reset_plus_step_ms': cycle_ms,
                       'episodes_per_s_reset_plus_step': B * world / (cycle_ms * 1e-3),
                       'min_relative_pivot': min_pivot,
                       'tolerance_pu': env.solve_opts.tol, 'byte_model': bm, 'lds_model': lm,
                       'kernel_launch': {k: ki[k] for k in ('waves_per_instance', 'lds_bytes_per_instance', 'instances_per_cu')},
                       'shared_slots': env.plan.info['n_shared'],      # fill blocks living in the LDS slot of a dead block (plan.cpp share_slots)
                       'device': {'compute_units': n_cu, 'simds': n_simd}},
            'roofline': roof,
        }
        full = world == 1 and not args.no_cpu_baseline
        if full:
            # the second half of BASELINE.json's metric, on 64 instances (8 for the N-1 configuration: 251 solves each)
            n_check = min(8 if args.config == 5 else 64, B)
            out['config']['max_abs_v_err_pu'] = voltage_check(env, args.config, n_check)
            out['config']['max_abs_v_err_against'] = (f'CPU oracle (oracle/pf_oracle.py), base-case |V| of {n_check} instances; '
                                                      'pandapower is not installed here')
        if full and not args.no_also and args.batch is None and args.as_rank is None:
            # the other BASELINE configurations on this GPU, briefly (>= 5 steps; the N-1 configuration 3), after the
            # headline measurement and before the CPU baseline
            env.close()
            del env
            torch.cuda.empty_cache()
            out['also'] = {}
            for c in sorted(CONFIGS):
                if c == args.config:
                    continue
                try:
                    out['also'][f'config{c}'] = also_config(c, device, 3 if c == 5 else 20, 1 if c == 5 else 5)       # (20 launches after 5: a run of 5 still sees the clock ramp, config 3 +4 %)
                except Exception as exc:           # (a failure here must not cost the headline line)
                    out['also'][f'config{c}'] = {'error': f'{type(exc).__name__}: {exc}'}
        if full and not args.no_also and args.batch is None and args.as_rank is None:
            # the SAME workloads at the reference's own solver settings: pandapower's init='auto' — a DC
            # power flow first on every grid fed above 70 kV, which is all of these — and, for the N-1
            # configuration, every contingency solved from scratch (security_constrained.py:53).  Same fixed point and
            # tolerance as the headline; the iteration path, and with it `iterations` and the time, are the reference's.
            out['reference_settings'] = {
                'what': "BatchedOpfEnv(reference_faithful=True): init='auto', contingency_start='flat', enforce_q_lims on pypower's own "
                        'path (generators with min_q = max_q start as PV buses: config 3), carry_over_state=True (concerns reset '
                        'only); the headline and `also` run init=flat / contingency_start=base_case / such generators pinned from the start'}
            for c in (2, 3, 5):
                try:
                    out['reference_settings'][f'config{c}'] = also_config(c, device, 3 if c == 5 else 20, 1 if c == 5 else 5,
                                                                          reference_faithful=True)
                except Exception as exc:
                    out['reference_settings'][f'config{c}'] = {'error': f'{type(exc).__name__}: {exc}'}
        # timed after the GPU work, in child processes (one, then one per host core)
        out['cpu_baseline'] = cpu_baseline(args.config) if full else None
        print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
