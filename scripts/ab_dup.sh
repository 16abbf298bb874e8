#!/bin/bash
# Cost of each prologue / epilogue phase of k_step in the real mix: developer builds (-DOPFX_DEV_MIN: headline instantiations
# only, seconds to compile) that run phase k TWICE (-DOPFX_DUP=1<<k), timed against the build that runs every phase once.
#   scripts/ab_dup.sh build            (here, no GPU)   -> opfgym_amd/libopfx_dup<k>.so, k = none 0 .. 11
#   scripts/ab_dup.sh run [rounds]     (on the GPU box) -> one line per build, bench workload of config 2
names=(stage_row actions table_obs injections cost_pre init_voltage compute_results derived_rows constraints cost_post results_out obs_out)
if [ "$1" = "build" ]; then
  F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -disable-machine-licm -Iinclude -Iopfgym_amd/csrc -DOPFX_DEV_MIN"
  hipcc $F -o opfgym_amd/libopfx_dupnone.so opfgym_amd/csrc/plan.cpp opfgym_amd/csrc/opfx.hip 2>/dev/null &
  for k in $(seq 0 11); do
    hipcc $F -DOPFX_DUP=$((1<<k)) -o opfgym_amd/libopfx_dup$k.so opfgym_amd/csrc/plan.cpp opfgym_amd/csrc/opfx.hip 2>/dev/null &
    if [ $((k % 4)) = 3 ]; then wait; fi
  done
  wait; ls opfgym_amd/libopfx_dup*.so | wc -l
else
  n=${2:-2}
  for i in $(seq $n); do
    echo "none: $(OPFX_LIB=opfgym_amd/libopfx_dupnone.so python scripts/probe_step.py 8192 2 2>&1 | grep k_step | tail -1)"
    for k in $(seq 0 11); do
      echo "${names[$k]}: $(OPFX_LIB=opfgym_amd/libopfx_dup$k.so python scripts/probe_step.py 8192 2 2>&1 | grep k_step | tail -1)"
    done
  done
fi
