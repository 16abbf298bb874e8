"""Developer probe: duration of one reset launch against the batch size (HIP events) — tells a latency-bound reset
(time steps up with the number of row generations per CU) from a bandwidth-bound one (time proportional to rows)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
from opfgym_amd import envs
from opfgym_amd import capi
capi.set_default_debug(capi.debug_from_env())      # this harness is steered through OPFX_* variables (see capi.debug_from_env)
for B in (64, 256, 1024, 2048, 4096, 8192, 16384, 32768):
    env = envs.VoltageControl(simbench_network_name='1-MV-urban--0-sw', batch_size=B, device='cuda:0', seed=0)
    for _ in range(3):
        env.reset()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        env.reset()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f'B={B:6d}  reset {ms * 1e3:8.1f} us   {ms * 1e6 / B:7.2f} ns per row   {B * (env.nx + env.n_obs_raw) * 8 / (ms * 1e-3) / 1e12:6.2f} TB/s written')
    env.close()
