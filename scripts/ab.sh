#!/bin/bash
# A/B timing of several builds of libopfx on the SAME GPU box (devices differ by several percent):
#   scripts/ab.sh [-n rounds] libA.so libB.so [libC.so ...]
n=3
if [ "$1" = "-n" ]; then n=$2; shift 2; fi
for i in $(seq $n); do
  for lib in "$@"; do
    echo "== $lib"
    OPFX_LIB=$lib python scripts/quick_bench_solve.py 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-80
    OPFX_LIB=$lib python scripts/probe_step.py 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
