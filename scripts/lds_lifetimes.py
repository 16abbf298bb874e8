"""Plan-level estimate (CPU): how much LDS a lifetime-based slot allocation of the LOWER blocks would free.

A block (i, j) whose column j is eliminated before its row i is read for the last time at level(j) (as the multiplier A_ik of
that level); upper and diagonal blocks live until the back substitution.  Peak of the live lower blocks over the levels against
their total = what slot sharing could save at best (full Newton only: chord steps re-read the lower blocks).

    python scripts/lds_lifetimes.py   ->  profiles/r04_lds_lifetimes.txt
"""
import sys; import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import numpy as np
from opfgym_amd import capi, grids
from opfgym_amd.case import net_to_case
capi.set_default_debug(capi.debug_from_env())      # this harness is steered through OPFX_* variables (see capi.debug_from_env)
from plan_emulator import load_plan
for code in ('1-HV-mixed--0-sw','1-HV-urban--0-sw','1-MV-urban--0-sw'):
    case = net_to_case(grids.get_grid(code)[0]); plan = capi.Plan(case); P = load_plan(plan); info = plan.info
    nlev = info['n_levels']; piv = P['piv_bus']; lev_of = {}
    for lev in range(nlev):
        for k in piv[P['lev_pptr'][lev]:P['lev_pptr'][lev+1]]: lev_of[int(k)] = lev
    br, bc = P['blk_row'], P['blk_col']
    nblk = info['n_blk']; nfull = info['n_full']
    birth = np.full(nblk, -1)
    tgt = P['tgt_blk']; 
    for lev in range(nlev):
        for t in range(P['lev_tptr'][lev], P['lev_tptr'][lev+1]):
            b = tgt[t]
            if b >= 0 and b in set(P['fill_blk'].tolist()) and birth[b] == -1: pass
    fill = set(int(b) for b in P['fill_blk'])
    first_t = {}
    for lev in range(nlev):
        for t in range(P['lev_tptr'][lev], P['lev_tptr'][lev+1]):
            b = int(tgt[t])
            if b >= 0 and b not in first_t: first_t[b] = lev
    INF = 10**9
    events = []
    n_lower = 0; dbl_lower = 0
    for b in range(nblk):
        i, j = int(br[b]), int(bc[b])
        if i == j or i not in lev_of or j not in lev_of: continue
        # lower: column j eliminated before row i  (block (i,j) is multiplier A_ik with k=j)
        if lev_of[j] < lev_of[i]:
            size = 4 if b < nfull else 2
            bl = first_t.get(b, -1) if b in fill else -1
            dl = lev_of[j]
            n_lower += 1; dbl_lower += size
            events.append((bl, dl, size))
    # peak of live doubles among lower blocks over levels (-1 = phase A)
    peak = 0
    for lev in range(-1, nlev):
        live = sum(sz for bl, dl, sz in events if bl <= lev <= dl)
        peak = max(peak, live)
    print(code, 'lds_doubles', info['lds_doubles'], 'lower blocks', n_lower, 'doubles', dbl_lower, 'peak live doubles', peak, 'saving KB', (dbl_lower-peak)*8/1024, 'new KB', (info['lds_doubles']-(dbl_lower-peak))*8/1024)
