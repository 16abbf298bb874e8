"""Developer probe (stamps build): where a reset row spends its cycles.
    python __graft_entry__.py --stamps;  OPFX_LIB=opfgym_amd/libopfx_stamps.so OPFX_STAMPS=1 python scripts/probe_reset_stamps.py [config]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
import torch
from opfgym_amd import capi, envs
capi.set_default_debug(capi.debug_from_env())      # this harness is steered through OPFX_* variables (see capi.debug_from_env)
import bench
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cls, kw, B, _, _ = bench.CONFIGS[cfg]
B = min(B, 8192)
env = getattr(envs, cls)(batch_size=B, device='cuda:0', seed=0, **kw)
for _ in range(3):
    env.reset()
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 32)()
capi.lib().opfx_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
capi.lib().opfx_debug_read_stamps(env.ctx.handle, out)
n = 10
for _ in range(n):
    env.reset()
capi.lib().opfx_debug_read_stamps(env.ctx.handle, out)
names = ['time step', 'template', 'profile tables', 'vector ops', 'initial action', 'observation', 'store x (drained)']
tot = sum(out[:7])
for k, nm in enumerate(names):
    print(f'  {nm:22s} {out[k] // n:9d} cycles per reset  {100 * out[k] / tot:5.1f}%')
for k, nm in ((8, 'ops: stage pointers'), (9, 'ops: barrier'), (10, 'ops: chunk descriptors + constants (drained)'), (11, 'ops: LDS reads, arithmetic, LDS writes (drained)')):
    if out[k]:
        print(f'    {nm:48s} {out[k] // n:9d} cycles per reset')
print(f'  total {tot // n} cycles per reset for the rows of wavefront 0 of workgroup 0 (nx={env.nx}, na={env.n_actions}, n_ops={len(env.ops.ops)}, tables={len(env.tables)})')
