"""Timing of opfx_solve on a grid past the LDS (memory-resident wave-team kernel) — developer script, GPU box.

    python scripts/bench_large_grid.py [nb] [batch] [reps]

Prints one JSON line: solves/s, mean NR iterations, and the SURVEY §8d algorithmic bytes per solve
(it * 8 * (2 nnzJ + 2 nnzLU + 4 nJ + 4 nb) + I/O) against the kernel time — for this kernel the model describes what
really happens: the LU block values stream through the memory hierarchy (L2 / Infinity Cache / HBM) every phase."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import numpy as np  # noqa: E402
import torch  # noqa: E402

from helpers import random_injections  # noqa: E402
from opfgym_amd import capi, grids  # noqa: E402
from opfgym_amd.case import net_to_case  # noqa: E402
capi.set_default_debug(capi.debug_from_env())      # this harness is steered through OPFX_* variables (see capi.debug_from_env)

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
net, _ = grids.synthetic_hv(5, nb=nb, n_ext=2, n_gen=10)
case = net_to_case(net)
plan = capi.Plan(case)
info = plan.info
ctx = capi.Context(plan, 0)
p, q = random_injections(net, case, B, seed=3, lo=0.5, hi=1.0)
dev = torch.device('cuda:0')
pt, qt = torch.tensor(p, device=dev), torch.tensor(q, device=dev)
out = capi.solve(ctx, pt, qt)
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(reps):
    out = capi.solve(ctx, pt, qt)
ev1.record()
torch.cuda.synchronize()
ms = ev0.elapsed_time(ev1) / reps
it = float(out['iterations'].double().mean())
n_j = info['npv'] + 2 * info['npq']
per_it = 8 * (2 * info['nnz_j'] + 2 * 4 * info['n_blk'] + 4 * n_j + 4 * info['nb'])
io = 8 * (2 * info['nb'] + 2 * info['nb'] + info['nbr'] + 2 * info['nref'] + 3)
b_solve = io + it * per_it
print(json.dumps({'nb': info['nb'], 'n_blk': info['n_blk'], 'levels': info['n_levels'], 'batch': B, 'kernel_ms': ms,
                  'solves_per_s': B / (ms * 1e-3), 'mean_nr_iterations': it, 'converged': float(out['converged'].double().mean()),
                  'lds_resident_bytes_needed': info['lds_doubles'] * 8, 'algorithmic_bytes_per_solve': b_solve,
                  'algorithmic_GBps': b_solve * B / (ms * 1e-3) / 1e9, 'frac_of_8TBps': b_solve * B / (ms * 1e-3) / 8e12}))
