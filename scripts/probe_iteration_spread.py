import sys, os; sys.path[:0]=['/root/repo']
import numpy as np, torch, heapq
import bench
from opfgym_amd import envs
from opfgym_amd import capi
capi.set_default_debug(capi.debug_from_env())      # this harness is steered through OPFX_* variables (see capi.debug_from_env)
for cfg in (2, 3, 4):
    cls, kw, B, _, _ = bench.CONFIGS[cfg]
    env = getattr(envs, cls)(batch_size=B, device='cuda:0', seed=0, **kw)
    env.reset()
    a = torch.rand(B, env.n_actions, dtype=torch.float64, device='cuda:0')
    out = env.step(a)
    info = out[4]
    it = info.get('total_iterations', info['iterations']).cpu().numpy().astype(int)
    vals, cnt = np.unique(it, return_counts=True)
    print('config', cfg, 'B', B, dict(zip(vals.tolist(), cnt.tolist())), 'mean', it.mean())
    teams = env.kernel_info()
    print('  kernel_info', teams)
    nslots = 256 * teams['instances_per_cu']
    # time model: t = base + c*it  with Newton share 0.75 at the mean
    c = 0.75 / it.mean(); base = 0.25
    t = base + c * it
    static = max(t[s::nslots].sum() for s in range(nslots))
    h = [0.0] * nslots; heapq.heapify(h)
    for x in t:
        heapq.heappush(h, heapq.heappop(h) + x)
    print('  slots', nslots, 'per slot', B / nslots, 'static makespan', round(static, 3), 'greedy', round(max(h), 3), 'mean load', round(t.sum() / nslots, 3))
    env.close()
