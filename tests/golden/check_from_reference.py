"""Container-only check (needs /root/reference): `BatchedOpfEnv.from_reference(ref_env)` on LIVE reference
environments — built from the reference's own classes under the stub packages — defines the same problem as
the `opfgym_amd.envs` class that stands for it (which reads the recorded definition): identical element
tables, keys, constraints and reward parameters.  Prints one line per class; exits non-zero on a difference.

    PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/check_from_reference.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path[:0] = [ROOT, HERE]

import numpy as np  # noqa: E402

from opfgym_amd import definition, envs  # noqa: E402
from opfgym_amd.batched_env import BatchedOpfEnv  # noqa: E402
from scenarios import SCENARIOS  # noqa: E402

# 1) recorded side first (the reference is not importable yet)
recorded = {}
for name in ('vc_mv_small', 'qm_mv_small', 'eco_hv_small', 'maxren_lv', 'loadshed_mv_small', 'sc_hv_small',
             'vc_parameterized', 'vc_resobs_diff', 'multistage_lv', 'partial_obs_lv', 'mixed_lv'):
    cls, kw, _, seed = SCENARIOS[name]
    recorded[name] = getattr(envs, cls)(batch_size=1, defer_device=True, seed=seed, **kw)

# 2) live side
sys.path[:0] = [os.path.join(HERE, '_stubs'), '/root/reference']
import opfgym.envs  # noqa: E402
import opfgym.examples.security_constrained as ex_sc  # noqa: E402
import opfgym.examples.multi_stage as ex_ms  # noqa: E402
import opfgym.examples.partial_obs as ex_po  # noqa: E402
import opfgym.examples.mixed_continuous_discrete as ex_mcd  # noqa: E402

REF = {'VoltageControl': opfgym.envs.VoltageControl, 'QMarket': opfgym.envs.QMarket, 'EcoDispatch': opfgym.envs.EcoDispatch,
       'MaxRenewable': opfgym.envs.MaxRenewable, 'LoadShedding': opfgym.envs.LoadShedding,
       'SecurityConstrained': ex_sc.SecurityConstrained, 'MultiStageOpf': ex_ms.MultiStageOpf,
       'PartiallyObservable': ex_po.PartiallyObservable, 'MixedContinuousDiscrete': ex_mcd.MixedContinuousDiscrete}
import logging  # noqa: E402


class _Notices(logging.Handler):
    def __init__(self):
        super().__init__()
        self.lines = []

    def emit(self, record):
        self.lines.append(record.getMessage())


notices = _Notices()
logging.getLogger('opfgym_amd').addHandler(notices)
bad = 0
for name, rec in recorded.items():
    cls, kw, _, seed = SCENARIOS[name]
    ref_env = REF[cls](seed=seed, **kw)
    live = BatchedOpfEnv.from_reference(ref_env, defer_device=True, seed=seed)
    ok = live.n_actions == rec.n_actions and live.store.n == rec.store.n
    ok &= [(u, c, list(i)) for u, c, i in live.obs_keys] == [(u, c, list(i)) for u, c, i in rec.obs_keys]
    ok &= [(u, c, list(i)) for u, c, i in live.act_keys] == [(u, c, list(i)) for u, c, i in rec.act_keys]
    ok &= np.array_equal(live.store.row_template(), rec.store.row_template(), equal_nan=True)
    ok &= [type(c_).__name__ for c_ in live.constraints] == [type(c_).__name__ for c_ in rec.constraints]
    ok &= len(live.ops.ops) == len(rec.ops.ops) and all(a[:4] == b[:4] for a, b in zip(live.ops.ops, rec.ops.ops))
    lr, rr = live.host_definition()['reward_function'], rec.host_definition()['reward_function']
    ok &= type(lr).__name__ == type(rr).__name__ and lr.scaling_params == rr.scaling_params and lr.penalty_weight == rr.penalty_weight
    ok &= np.array_equal(live.train_steps, rec.train_steps)
    ok &= tuple((u, c, list(i)) for u, c, i in live.n_minus_one_keys) == tuple((u, c, list(i)) for u, c, i in rec.n_minus_one_keys)
    checks = dict(n=live.n_actions == rec.n_actions and live.store.n == rec.store.n,
                  obs=[(u, c, list(i)) for u, c, i in live.obs_keys] == [(u, c, list(i)) for u, c, i in rec.obs_keys],
                  tmpl=np.array_equal(live.store.row_template(), rec.store.row_template(), equal_nan=True),
                  cons=[type(c_).__name__ for c_ in live.constraints] == [type(c_).__name__ for c_ in rec.constraints],
                  ops=len(live.ops.ops) == len(rec.ops.ops), rew=(type(lr).__name__, lr.scaling_params, lr.penalty_weight) == (type(rr).__name__, rr.scaling_params, rr.penalty_weight),
                  split=np.array_equal(live.train_steps, rec.train_steps))
    if not ok:
        print('   ', {k: v for k, v in checks.items() if not v}, live.store.n, rec.store.n)
    print(f'{name:22s} {"same definition" if ok else "DIFFERENT"}  ({cls}, {live.n_actions} actions, {len(live.ops.ops)} reset ops)')
    bad += not ok
# 3) the solver settings `from_reference` does NOT take from the reference by default are said out loud, once per setting,
#    and `reference_faithful=True` leaves nothing to say
cls, kw, _, seed = SCENARIOS['sc_hv_small']
ref_env = REF[cls](seed=seed, **kw)
text = ' '.join(notices.lines)
said = all(w in text for w in ("init='flat'", "contingency_start='base_case'", 'carry_over_state=False', 'reference_faithful=True'))
n_before = len(notices.lines)
fast = BatchedOpfEnv.from_reference(ref_env, defer_device=True, seed=seed)                   # (same settings again: no second notice)
faithful = BatchedOpfEnv.from_reference(ref_env, defer_device=True, seed=seed, reference_faithful=True)
quiet = len(notices.lines) == n_before
ok = said and quiet and set(fast.reference_deviations) == {'init', 'contingency_start', 'carry_over_state', 'pin_point_q_ranges'} \
    and faithful.reference_deviations == {} and faithful.init == 'dc' and faithful.solve_opts.contingency_start == 1 \
    and faithful.carry_over_state and fast.init == 'flat' and fast.solve_opts.contingency_start == 0
print(f'{"solver-settings notice":22s} {"same definition" if ok else "DIFFERENT"}  ({len(notices.lines)} notices; said={said}, quiet={quiet})')
bad += not ok
sys.exit(1 if bad else 0)
