"""Developer probe: k_step kernel time (HIP events) for the bench workload."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
from opfgym_amd import capi, envs
capi.set_default_debug(capi.debug_from_env())      # this harness is steered through OPFX_* variables (see capi.debug_from_env)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
CFG = int(sys.argv[2]) if len(sys.argv) > 2 else 2
import bench
cls, kw, _, _, _ = bench.CONFIGS[CFG]
if os.environ.get('OPFX_REF'):                      # the reference's own solver settings (DC start, contingencies from scratch)
    kw = dict(kw, reference_faithful=True)
env = getattr(envs, cls)(batch_size=B, device='cuda:0', seed=0, **kw)
rng = np.random.default_rng(1234)
env.reset(options={'step': rng.choice(env.train_steps, B)})
actions = torch.as_tensor(rng.random((B, env.n_actions)), device='cuda:0')
if os.environ.get('OPFX_FIXED_IT'):
    env.solve_opts.tol = 0.0; env.solve_opts.max_iter = int(os.environ['OPFX_FIXED_IT'])
for _ in range(3):
    env.step(actions)
io = env._io(actions, False)
ms = capi.C.c_float()
capi.check(capi.lib().opfx_time_steps(env._env_handle, B, capi.C.byref(io), capi.C.byref(env.solve_opts), 20,
                                      capi._stream(), capi.C.byref(ms)))
print(f'k_step {ms.value/20:.4f} ms  it={env.buf["iterations"].double().mean().item():.2f}')
if os.environ.get('OPFX_STAMPS'):
    import ctypes
    out = (ctypes.c_ulonglong * 32)()
    capi.lib().opfx_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    capi.lib().opfx_debug_read_stamps(env.ctx.handle, out)      # clear warm-up sums
    env.step(actions)
    capi.lib().opfx_debug_read_stamps(env.ctx.handle, out)
    names = {0: ' prologue: cost rows (rest)', 1: ' A: mods + norm reduction', 2: 'NR phase B', 3: 'NR phase C', 4: 'NR phase D',
             5: 'solve_instance total(excl.)', 6: 'compute_results', 7: 'constraints', 8: 'objective+results out',
             9: 'obs out', 20: ' team: tail chain + barrier', 10: ' A: desc prefetch + fill zero', 11: ' A: overflow entries', 12: ' A: bus rounds', 15: 'loop top/reward/outputs', 16: ' prologue: stage row', 17: ' prologue: actions', 18: ' prologue: table obs', 19: ' prologue: injections'}
    tot = sum(out)
    n_inst = (B + 1535) // 1536
    for k in sorted(names):
        print(f'  {names[k]:34s} {out[k]:10d} cyc  {100*out[k]/tot:5.1f}%')
    print(f'  total {tot} cycles for ~{n_inst} instances of workgroup 0 (100 MHz ticks? see s_memtime)')
