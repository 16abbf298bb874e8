"""Developer probe (stamps build): when each workgroup of a step launch finishes, against the Newton iterations it ran.
    python __graft_entry__.py --stamps;  OPFX_LIB=opfgym_amd/libopfx_stamps.so OPFX_STAMPS=1 python scripts/probe_finish_times.py [config]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np, torch
from opfgym_amd import capi, envs
capi.set_default_debug(capi.debug_from_env())      # this harness is steered through OPFX_* variables (see capi.debug_from_env)
import bench
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cls, kw, B, _, _ = bench.CONFIGS[cfg]
env = getattr(envs, cls)(batch_size=B, device='cuda:0', seed=0, **kw)
env.reset(); env.step(torch.rand(B, env.n_actions, dtype=torch.float64, device='cuda:0')); torch.cuda.synchronize()
info = env.kernel_info()
n_wg = 256 * info['instances_per_cu']
lib = capi.lib()
lib.opfx_debug_read_finish.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
out = np.zeros((n_wg, 6))
a = torch.rand(B, env.n_actions, dtype=torch.float64, device='cuda:0')
for rep in range(4):
    env.reset()
    torch.cuda.synchronize()
    lib.opfx_debug_read_finish(env.ctx.handle, out.ctypes.data, n_wg)      # clear
    env.step(a)
    torch.cuda.synchronize()
    lib.opfx_debug_read_finish(env.ctx.handle, out.ctypes.data, n_wg)
    t = (out[:, 0] - out[:, 0].min()) / 100.0      # us (100 MHz wall clock)
    n, it = out[:, 1], out[:, 2]
    print(f'launch {rep}: last finish - first finish = {t.max():.1f} us; finish time percentiles 10/50/90/99/100: '
          + ' '.join(f'{np.percentile(t, q):.1f}' for q in (10, 50, 90, 99, 100)))
    for k in sorted(set(it.astype(int))):
        m = it == k
        print(f'   workgroups with {k:3d} iterations in {n[m].mean():.1f} instances: {m.sum():5d}, finish {t[m].mean():6.1f} us mean, {t[m].min():6.1f} .. {t[m].max():6.1f}')
    st = (out[:, 3] - out[:, 3].min()) / 100.0
    dur = (out[:, 0] - out[:, 3]) / 100.0
    print(f'   start skew: last start - first start = {st.max():.1f} us (percentiles 50/90/99: ' + ' '.join(f'{np.percentile(st, q):.1f}' for q in (50, 90, 99))
          + f'); busy time per workgroup: mean {dur.mean():.1f} us, std {dur.std():.1f}, min {dur.min():.1f}, max {dur.max():.1f}')
    for k in sorted(set(it.astype(int))):
        m = it == k
        print(f'      {k:3d} iterations: busy {dur[m].mean():6.1f} us mean, std {dur[m].std():4.1f}, {dur[m].min():6.1f} .. {dur[m].max():6.1f}')
    hw = out[:, 4].astype(np.int64); xcc = out[:, 5].astype(np.int64) & 0xF
    simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    cu_key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    resid = dur - np.array([dur[it == k].mean() for k in it])        # busy time minus the mean of its iteration class
    keys = np.unique(cu_key)
    cu_mean = np.array([resid[cu_key == k].mean() for k in keys]); cu_n = np.array([(cu_key == k).sum() for k in keys])
    print(f'   {len(keys)} CUs seen, workgroups per CU {cu_n.min()}..{cu_n.max()}; residual busy time: overall std {resid.std():.1f} us; std of the CU means {cu_mean.std():.1f} us '
          f'(min {cu_mean.min():.1f}, max {cu_mean.max():.1f}); std within CUs {np.sqrt(np.mean([resid[cu_key == k].var() for k in keys])):.1f} us')
    print('   residual by XCC:', ' '.join(f'{resid[xcc == x].mean():+.1f}' for x in range(8)), '| by SIMD:', ' '.join(f'{resid[simd == x].mean():+.1f}' for x in range(4)))
    xcd = np.arange(n_wg) % 8
    print('   mean finish by XCD (workgroup number mod 8):', ' '.join(f'{t[xcd == x].mean():.1f}' for x in range(8)))
