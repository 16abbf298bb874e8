"""Developer probe (diagnostic build only: python __graft_entry__.py --probe, then
OPFX_LIB=opfgym_amd/libopfx_probe.so python scripts/probe_round.py): cycle-counter ticks per segment of a
factor item of k_solve (workgroup 0) and LDS microbenchmarks, for one lone instance and for a full batch."""
import sys, os, ctypes
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
from opfgym_amd import capi, grids
from opfgym_amd.case import net_to_case
from helpers import random_injections
net, _ = grids.get_grid('1-MV-urban--0-sw')
case = net_to_case(net); plan = capi.Plan(case); ctx = capi.Context(plan, 0)
for B in (1, 8192):
    p, q = random_injections(net, case, B, 1)
    pt, qt = torch.tensor(p, device='cuda:0'), torch.tensor(q, device='cuda:0')
    out = (ctypes.c_ulonglong * 16)()
    for _ in range(2): capi.solve(ctx, pt, qt)
    torch.cuda.synchronize()
    capi.lib().opfx_debug_read_probe(out)
    capi.solve(ctx, pt, qt); torch.cuda.synchronize()
    capi.lib().opfx_debug_read_probe(out)
    n = max(out[5], 1)
    m = max(out[9], 1)
    print(f'   calib: 256 dependent fma {out[6]/m:.0f} ticks, 16 independent ds_read_b64 {out[7]/m:.0f}, 16 dependent LDS reads {out[8]/m:.0f}')
    print(f'   16x: b128 linear {out[10]/m:.0f}, b64 random {out[11]/m:.0f}, b128 random {out[12]/m:.0f}, b64 linear {out[13]/m:.0f}')
    print(f'B={B}: rounds={out[5]}  desc-wait {out[0]/n:.0f}  lds-reads {out[1]/n:.0f}  fp {out[2]/n:.0f}  atomics {out[3]/n:.0f}  clk-overhead {out[4]/n:.0f}  (ticks per factor round, workgroup 0)')
