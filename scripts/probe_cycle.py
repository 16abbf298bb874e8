"""Where the host time of a reset + step cycle goes (developer probe, GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np, torch
from opfgym_amd import envs
from opfgym_amd import capi
capi.set_default_debug(capi.debug_from_env())      # this harness is steered through OPFX_* variables (see capi.debug_from_env)
B = 8192
env = envs.VoltageControl(simbench_network_name='1-MV-urban--0-sw', batch_size=B, device='cuda:0', seed=0)
act = torch.rand(B, env.n_actions, dtype=torch.float64, device='cuda:0')
def timed(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
print('reset            %.3f ms' % timed(lambda: env.reset()))
print('step             %.3f ms' % timed(lambda: env.step(act)))
print('reset+step       %.3f ms' % timed(lambda: (env.reset(), env.step(act))))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(50): env.reset()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
