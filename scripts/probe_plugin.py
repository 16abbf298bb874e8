"""Developer probe: end-to-end time of the batch-1 plug-in `power_flow_solver(net)` (opf_env.py:53 seam)."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opfgym_amd import grids, power_flow_solver
from opfgym_amd import capi
capi.set_default_debug(capi.debug_from_env())      # this harness is steered through OPFX_* variables (see capi.debug_from_env)
code = sys.argv[1] if len(sys.argv) > 1 else '1-MV-urban--0-sw'
net, _ = grids.get_grid(code)
for _ in range(3):
    power_flow_solver(net)
n = 30
t0 = time.perf_counter()
for _ in range(n):
    power_flow_solver(net)
dt = (time.perf_counter() - t0) / n
print(f'{code}: {dt*1e3:.2f} ms per power_flow_solver(net) call ({len(net.bus)} buses)')
if len(sys.argv) > 2:
    pr = cProfile.Profile(); pr.enable()
    for _ in range(10):
        power_flow_solver(net)
    pr.disable(); pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
