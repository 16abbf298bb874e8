"""Developer check: a long run of reset + step cycles — device and host memory stay flat, results stay finite."""
import os, sys, time, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opfgym_amd.vector_env import make_vec
from opfgym_amd import capi
capi.set_default_debug(capi.debug_from_env())      # this harness is steered through OPFX_* variables (see capi.debug_from_env)
B = 8192
vec = make_vec('VoltageControl-v0', B, simbench_network_name='1-MV-urban--0-sw', device='cuda:0', seed=0)
vec.reset(seed=1)
a = torch.rand(B, vec.single_action_space.shape[0], device='cuda:0', dtype=torch.float64)
def mem():
    return torch.cuda.memory_allocated() / 2**20, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024
for _ in range(200):
    vec.step(a)
torch.cuda.synchronize()
m0 = mem()
t0 = time.perf_counter()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
bad = 0
for k in range(n):
    obs, r, term, trunc, info = vec.step(a)
    if k % 500 == 0:                      # (also keeps the host from running far ahead of the device queue)
        bad += int((~torch.isfinite(r)).sum().item())
torch.cuda.synchronize()
dt = time.perf_counter() - t0
m1 = mem()
print(f'{n} cycles in {dt:.1f} s ({B*n/dt/1e6:.1f} M episodes/s); device MiB {m0[0]:.1f} -> {m1[0]:.1f}; host RSS MiB {m0[1]:.0f} -> {m1[1]:.0f}; non-finite rewards sampled: {bad}')
