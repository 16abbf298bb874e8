"""Developer probe (GPU box): the single-wave step kernel on a SMALL grid (33-bus `mv-small`: ~5 KB of LDS per instance), where
registers — not LDS — cap the resident wavefronts at two per SIMD.  Run once with the product library and once with a build
capped at 168 VGPRs (-DOPFX_MIN_WAVES_PER_SIMD=3: three wavefronts per SIMD):

    OPFX_LIB=opfgym_amd/libopfx_w3.so python scripts/probe_small_grid_occupancy.py [batch]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np  # noqa: E402
import torch  # noqa: E402

from opfgym_amd import envs  # noqa: E402
from opfgym_amd import capi
capi.set_default_debug(capi.debug_from_env())      # this harness is steered through OPFX_* variables (see capi.debug_from_env)

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
for code in ('mv-small', '1-LV-rural1--0-sw'):
    kw = dict(min_sgen_power=0.005, min_storage_power=0.005) if code.startswith('1-LV') else {}
    cls = envs.MaxRenewable if code.startswith('1-LV') else envs.VoltageControl
    env = cls(simbench_network_name=code, batch_size=B, device='cuda:0', seed=0, **kw)
    env.reset(seed=1)
    a = torch.as_tensor(np.random.default_rng(2).random((B, env.n_actions)), device='cuda:0')
    for _ in range(5):
        env.step(a)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(20):
            info = env.step(a)[4]
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 20)
    ki = env.kernel_info()
    print(f'{os.environ.get("OPFX_LIB", "libopfx.so")} {code} batch {B}: {best * 1e3:.4f} ms per step, {B / best / 1e6:.1f} M step/s, '
          f'resident per CU {ki["instances_per_cu"]}, lds {ki["lds_bytes_per_instance"]} B, converged {float(info["converged"].double().mean()):.3f}')
