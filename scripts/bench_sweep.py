"""Timing protocol of SURVEY.md §8(d) for the bench workload (VoltageControl, 144-bus MV grid):
batch sizes 1 .. 65536, >= 20 timed step() calls each on a persistent context, device time per
call from HIP events (median, p10, p90), end-to-end time of the synchronous Python call, and the
FP64 max-abs error of vm_pu / loading_percent / reward against the CPU oracle on the same inputs
(first instances of each batch).  One JSON line per batch size.  Developer/report script; the
contract benchmark is bench.py."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
import numpy as np, torch
from opfgym_amd import envs
from opfgym_amd import capi
capi.set_default_debug(capi.debug_from_env())      # this harness is steered through OPFX_* variables (see capi.debug_from_env)
from env_cases import oracle_env, product_env

GRID = '1-MV-urban--0-sw'
sizes = [int(a) for a in sys.argv[1:]] or [1, 64, 1024, 8192, 65536]
orc = oracle_env('vc_mv_urban', product_env('vc_mv_urban', defer_device=True))
for B in sizes:
    env = envs.VoltageControl(simbench_network_name=GRID, batch_size=B, device='cuda:0', seed=0)
    rng = np.random.default_rng(1234)
    steps = rng.choice(env.train_steps, B)
    env.reset(options={'step': steps})
    actions_np = np.random.default_rng(4321).random((B, env.n_actions))
    actions = torch.as_tensor(actions_np, device='cuda:0')
    for _ in range(3):
        out = env.step(actions)
    torch.cuda.synchronize()
    dev_ms, e2e_ms = [], []
    for _ in range(30):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        out = env.step(actions)
        e1.record()
        torch.cuda.synchronize()
        e2e_ms.append((time.perf_counter() - t0) * 1e3)
        dev_ms.append(e0.elapsed_time(e1))
    obs, reward, term, trunc, info = out
    n_chk = min(B, 8)
    err_v = err_l = err_r = 0.0
    vm = env.result_table('bus', 'vm_pu')[:n_chk].cpu().numpy()
    ld = env.result_table('line', 'loading_percent')[:n_chk].cpu().numpy()
    for k in range(n_chk):
        orc.reset(int(steps[k]))
        ref = orc.step(actions_np[k])
        err_v = max(err_v, float(np.abs(vm[k] - ref['vm_pu']).max()))
        err_l = max(err_l, float(np.abs(ld[k] - ref['line_loading']).max()))
        err_r = max(err_r, abs(float(reward[k]) - ref['reward']))
    q = lambda a, p: float(np.percentile(a, p))
    print(json.dumps({'batch': B, 'device_ms': {'p10': q(dev_ms, 10), 'median': q(dev_ms, 50), 'p90': q(dev_ms, 90)},
                      'end_to_end_ms': {'p10': q(e2e_ms, 10), 'median': q(e2e_ms, 50), 'p90': q(e2e_ms, 90)},
                      'env_steps_per_s_device': B / q(dev_ms, 50) * 1e3,
                      'env_steps_per_s_end_to_end': B / q(e2e_ms, 50) * 1e3,
                      'converged': float(info['converged'].double().mean()),
                      'max_abs_err_vs_cpu_oracle': {'vm_pu': err_v, 'line_loading_percent': err_l, 'reward': err_r,
                                                    'instances_checked': n_chk}}))
    env.close()
