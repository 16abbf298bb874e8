#!/bin/bash
# A/B timing of ONE build under two environments on the same GPU box (plan-time switches such as OPFX_PLAN_NO_BANK):
#   scripts/ab_env.sh "<configs>" "VAR=value" ["VAR2=value" ...]      ('' = the default environment)
cfgs=$1; shift
for i in 1 2; do
  for e in "$@"; do
    for c in $cfgs; do
      st=20; [ $c = 5 ] && st=2
      env $e python bench.py --config $c --steps $st --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$e]', 'config', d['config']['baseline_config'], 'kernel_ms %.4f' % d['roofline']['kernel_ms'], 'it %.3f' % d['config']['mean_nr_iterations_all_solves'], 'conv %.3f' % d['config']['converged_fraction'])"
    done
  done
done
