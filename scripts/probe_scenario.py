"""Developer probe: kernel time of one golden scenario's environment (tests/golden/scenarios.py) at a batch size, random actions.
    python scripts/probe_scenario.py <scenario> [batch] [steps]        (OPFX_LIB=... for A/B builds, OPFX_REF=1: reference_faithful)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
from env_cases import product_env
from opfgym_amd import capi
capi.set_default_debug(capi.debug_from_env())      # this harness is steered through OPFX_* variables (see capi.debug_from_env)
name = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
kw = dict(reference_faithful=True) if os.environ.get('OPFX_REF') else {}
env = product_env(name, batch_size=B, **kw)
rng = np.random.default_rng(7)
env.reset(options={'step': rng.choice(env.train_steps, B)})
actions = torch.as_tensor(rng.random((B, env.n_actions)), device='cuda:0')
for _ in range(3):
    env.step(actions)
io = env._io(actions, False)
ms = capi.C.c_float()
capi.check(capi.lib().opfx_time_steps(env._env_handle, B, capi.C.byref(io), capi.C.byref(env.solve_opts), n, capi._stream(), capi.C.byref(ms)))
print(f'{name} B={B}: k_step {ms.value / n:.4f} ms  it={env.buf["iterations"].double().mean().item():.2f}  converged={env.buf["converged"].double().mean().item():.3f}  kernel={env.kernel_info()}')
