"""A/B on one box: BASELINE config 5 under the reference's own solver settings (reference_faithful=True: DC start of every
contingency) with the rank-1 DC start of the contingencies (default) and with a DC pass per contingency
(opfx_debug_opts.no_rank1_dc = 1, round 5's path).  python scripts/ab_rank1_dc.py [steps]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from opfgym_amd import capi  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for no_rank1 in (0, 1, 0, 1):
    capi.set_default_debug(dict(no_rank1_dc=no_rank1))
    r = bench.also_config(5, torch.device('cuda:0'), steps, 1, reference_faithful=True)
    print(json.dumps({'no_rank1_dc': no_rank1, 'ms_per_step': r['ms_per_step'], 'kernel_ms': r['kernel_ms'], 'kernel': r['kernel'],
                      'iterations_all_solves': r['mean_nr_iterations_all_solves'], 'converged': r['converged_fraction'],
                      'max_abs_v_err_pu': r['max_abs_v_err_pu']}), flush=True)
