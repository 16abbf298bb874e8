"""Quickstart: 8192 VoltageControl environments stepped at once on one MI355X.

    python examples/quickstart.py            # needs a GPU and the built library
                                             # (python -c "import __graft_entry__ as g; g.build()")

The classes mirror the reference's (`opfgym.envs.VoltageControl(...)`): same constructor
arguments plus `batch_size` and `device`; reset/step return torch tensors on the GPU with a
leading batch axis.  SimBench itself is not available offline, so the grid behind the SimBench
code is this repo's synthetic stand-in with the same element counts.
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from opfgym_amd.envs import VoltageControl
from opfgym_amd.vector_env import make_vec

B = 8192
env = VoltageControl(simbench_network_name='1-MV-urban--0-sw', batch_size=B, device='cuda:0', seed=0)
obs, _ = env.reset(seed=0)                               # [B, n_obs] float64 on the GPU
print('observation', tuple(obs.shape), 'actions per instance', env.n_actions)

actions = torch.rand(B, env.n_actions, device='cuda:0', dtype=torch.float64)
obs, reward, terminated, truncated, info = env.step(actions)      # ONE kernel launch
torch.cuda.synchronize()
print('mean reward %.4f, valid %.1f %%, converged %.1f %%, NR iterations %.2f' % (
    reward.mean().item(), 100 * info['valids'].all(dim=1).double().mean().item(),
    100 * info['converged'].double().mean().item(), info['iterations'].double().mean().item()))
print('bus voltages of instance 0:', env.result_table('bus', 'vm_pu')[0, :5].tolist(), '...')

# the same thing as a vector environment with autoreset (what an RL library drives)
vec = make_vec('VoltageControl-v0', B, simbench_network_name='1-MV-urban--0-sw', device='cuda:0', seed=0)
vec.reset(seed=0)
for _ in range(3):
    vec.step(actions)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 100
for _ in range(n):
    obs, reward, terminated, truncated, info = vec.step(actions)   # step + reset of the finished episodes
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print('%.3f ms per step+reset cycle of %d environments = %.1f M episodes/s' % (dt * 1e3, B, B / dt / 1e6))
