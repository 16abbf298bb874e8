"""Developer probe: time opfx_solve at a given batch size (not the bench contract)."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from opfgym_amd import capi, grids
from opfgym_amd.case import net_to_case
capi.set_default_debug(capi.debug_from_env())      # this harness is steered through OPFX_* variables (see capi.debug_from_env)
from helpers import random_injections
code = sys.argv[1] if len(sys.argv) > 1 else '1-MV-urban--0-sw'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
net, _ = grids.get_grid(code)
case = net_to_case(net)
plan = capi.Plan(case)
print(plan.info)
ctx = capi.Context(plan, 0)
p, q = random_injections(net, case, B, 1)
dev = torch.device('cuda:0')
pt, qt = torch.tensor(p, device=dev), torch.tensor(q, device=dev)
kw = {}
if os.environ.get('OUTAGE'):
    from helpers import non_bridge_branches
    cand = non_bridge_branches(case)
    kw['outage'] = torch.tensor(cand[np.arange(B) % len(cand)].astype(np.int32), device=dev)
for k_, cast in (('tol', float), ('max_iter', int)):
    if os.environ.get(k_.upper()):
        kw[k_] = cast(os.environ[k_.upper()])
for _ in range(3):
    out = capi.solve(ctx, pt, qt, **kw)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 20
e0.record()
for _ in range(reps):
    out = capi.solve(ctx, pt, qt, **kw)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
it = out['iterations'].float().mean().item()
print(f'{code} B={B}: {ms:.3f} ms/batch  -> {B/ms*1e3:.3e} solves/s, mean it {it:.2f}, conv {out["converged"].float().mean().item()}')
