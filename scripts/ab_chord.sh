#!/bin/bash
# Chord steps (opfx_solve_opts.jacobian_reuse_tol) against full Newton on ONE GPU box, same library:
#   scripts/ab_chord.sh "<configs>" "<thetas>"      e.g.  scripts/ab_chord.sh "2 3 5" "0 0.01 0.1 1"
cfgs=${1:-"2 3 5"}; thetas=${2:-"0 0.01 0.1 1"}
for i in 1 2; do
  for c in $cfgs; do
    for th in $thetas; do
      st=20; [ $c = 5 ] && st=2
      python bench.py --config $c --steps $st --warmup 2 --windows 3 --no-cpu-baseline --reuse-tol $th 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
c=d['config']
print('config', c['baseline_config'], 'theta $th', 'kernel_ms %.4f' % d['roofline']['kernel_ms'], 'iterations/solve %.3f' % (c['mean_nr_iterations_all_solves'] / c['solves_per_step']), 'converged %.4f' % c['converged_fraction'])"
    done
  done
done
