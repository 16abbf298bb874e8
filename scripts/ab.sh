#!/bin/bash
# A/B timing of two builds of libopfx on the SAME GPU box (devices differ by several percent):
#   scripts/ab.sh libA.so libB.so [rounds]
a=$1; b=$2; n=${3:-3}
for i in $(seq $n); do
  for lib in $a $b; do
    echo "== $lib"
    OPFX_LIB=$lib python scripts/quick_bench_solve.py 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-80
    OPFX_LIB=$lib python scripts/probe_step.py 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
