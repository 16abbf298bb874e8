#!/bin/bash
# A/B timing of several builds of libopfx on the SAME GPU box through bench.py (devices differ by several percent):
#   scripts/ab_bench.sh "<configs>" libA.so libB.so [...]      e.g.  scripts/ab_bench.sh "2 3" opfgym_amd/libopfx.so opfgym_amd/libopfx_x.so
# AB_ARGS="--init dc" adds bench.py arguments to every run.
cfgs=$1; shift
for i in 1 2; do
  for lib in "$@"; do
    for c in $cfgs; do
      st=20; [ $c = 5 ] && st=2
      OPFX_LIB=$lib python bench.py --config $c --steps $st --warmup 2 --windows 3 --no-cpu-baseline $AB_ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', 'config', d['config']['baseline_config'], 'kernel_ms %.4f' % d['roofline']['kernel_ms'], 'frac %.3f' % d['roofline']['frac'], 'it %.3f' % d['config']['mean_nr_iterations_all_solves'])"
    done
  done
done
