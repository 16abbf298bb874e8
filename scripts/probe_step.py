"""Developer probe: k_step kernel time (HIP events) for the bench workload."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
from opfgym_amd import capi, envs
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
env = envs.VoltageControl(simbench_network_name='1-MV-urban--0-sw', batch_size=B, device='cuda:0', seed=0)
rng = np.random.default_rng(1234)
env.reset(options={'step': rng.choice(env.train_steps, B)})
actions = torch.as_tensor(rng.random((B, env.n_actions)), device='cuda:0')
for _ in range(3):
    env.step(actions)
io = env._io(actions, False)
ms = capi.C.c_float()
capi.check(capi.lib().opfx_time_steps(env._env_handle, B, capi.C.byref(io), capi.C.byref(env.solve_opts), 20,
                                      capi._stream(), capi.C.byref(ms)))
print(f'k_step {ms.value/20:.4f} ms  it={env.buf["iterations"].double().mean().item():.2f}')
