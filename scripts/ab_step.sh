#!/bin/bash
# A/B of k_step (HIP-event time of 20 launches, scripts/probe_step.py) for several builds on the SAME box, interleaved:
#   scripts/ab_step.sh [-n rounds] "<batch> <config>" libA.so libB.so ...
n=3
if [ "$1" = "-n" ]; then n=$2; shift 2; fi
args=$1; shift
for i in $(seq $n); do
  for lib in "$@"; do
    echo "$lib: $(OPFX_LIB=$lib python scripts/probe_step.py $args 2>&1 | grep k_step | tail -1)"
  done
done
