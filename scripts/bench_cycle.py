"""Developer/report script: full RL cycle rate (reset + step per episode, single-step env) through the
vector-environment adapter — what a training loop sees, launches and device-side RNG included."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np, torch
from opfgym_amd.vector_env import make_vec
from opfgym_amd import capi
capi.set_default_debug(capi.debug_from_env())      # this harness is steered through OPFX_* variables (see capi.debug_from_env)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
vec = make_vec('VoltageControl-v0', B, simbench_network_name='1-MV-urban--0-sw', device='cuda:0', seed=0)
vec.reset(seed=1)
a = torch.rand(B, vec.single_action_space.shape[0], device='cuda:0', dtype=torch.float64)
for _ in range(5):
    vec.step(a)
torch.cuda.synchronize()
n = 50
t0 = time.perf_counter()
for _ in range(n):
    obs, r, term, trunc, info = vec.step(a)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f'B={B}: {dt*1e3:.3f} ms per step+autoreset cycle -> {B/dt:.3e} episodes/s (same_step autoreset)')
env = vec.env
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n):
    env.reset()
torch.cuda.synchronize(); dr = (time.perf_counter() - t0) / n
t0 = time.perf_counter()
for _ in range(n):
    env.step(a)
torch.cuda.synchronize(); ds = (time.perf_counter() - t0) / n
print(f'  reset alone {dr*1e3:.3f} ms, step alone {ds*1e3:.3f} ms')
