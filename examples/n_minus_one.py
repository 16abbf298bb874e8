"""N-1 security-constrained voltage control: 1024 instances of a 372-bus HV grid, every non-islanding line as a
contingency — 251 power flows per instance and step, all inside ONE kernel launch.

    python examples/n_minus_one.py           # needs a GPU and the built library

`SecurityConstrainedVoltageControl` composes the reference's `SecurityConstrainedOpfEnv`
(security_constrained.py) with the VoltageControl problem definition (BASELINE configuration 5).  The step's reward
carries the worst case over all contingencies (violations are accumulated over the cases, security_constrained.py:
44-66); `info` additionally tells how hard the solver had to work and how well conditioned the solves were.
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from opfgym_amd.envs import SecurityConstrainedVoltageControl

B = 1024
env = SecurityConstrainedVoltageControl(simbench_network_name='1-HV-urban--0-sw', n_minus_one_lines='all',
                                        batch_size=B, device='cuda:0', seed=0)
print('contingencies per step:', len(env.contingencies), '| buses:', env.case.nb, '| actions per instance:', env.n_actions)
env.reset(seed=0)
actions = torch.rand(B, env.n_actions, device='cuda:0', dtype=torch.float64)
env.step(actions)
torch.cuda.synchronize()
t0 = time.perf_counter()
obs, reward, terminated, truncated, info = env.step(actions)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
solves = B * (1 + len(env.contingencies))
print('%.1f ms per step of %d instances = %.2f M power-flow solves/s' % (dt * 1e3, B, solves / dt / 1e6))
print('N-1 secure instances: %.1f %%; Newton iterations per instance and step: %.0f (base case %.1f); '
      'smallest relative pivot: %.3f' % (100 * info['valids'].all(dim=1).double().mean().item(),
                                        info['total_iterations'].double().mean().item(),
                                        info['iterations'].double().mean().item(), info['min_pivot'].min().item()))
# the reference restarts pandapower from the flat start for every contingency; `contingency_start='flat'` does the same
# (identical iteration counts), the default starts each contingency from the base-case solution
