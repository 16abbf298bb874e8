#!/bin/bash
# Cost of each prologue / epilogue phase of k_step in the real mix: developer builds (the headline's kernel translation
# unit only, seconds to compile) that run phase k TWICE (-DOPFX_DUP=1<<k), timed against the build that runs every phase once.
#   scripts/ab_dup.sh build            (here, no GPU)   -> opfgym_amd/libopfx_dup<k>.so, k = none 0 .. 11
#   scripts/ab_dup.sh run [rounds]     (on the GPU box) -> one line per build, bench workload of config 2
names=(stage_row actions table_obs injections cost_pre init_voltage compute_results derived_rows constraints cost_post results_out obs_out)
if [ "$1" = "build" ]; then
  # (the headline's kernel translation unit alone — k_step<2,1,SPEC=3> lives in k_step_plain3.hip — plus k_solve / k_reset;
  #  the accessors of the units left out fall back to the weak ones of opfx.hip, __graft_entry__.build_variant)
  units="['k_step_plain3.hip', 'k_solve.hip', 'k_reset.hip']"
  python -c "import __graft_entry__ as g; g.build_variant('opfgym_amd/libopfx_dupnone.so', [], $units)"
  for k in $(seq 0 11); do
    python -c "import __graft_entry__ as g; g.build_variant('opfgym_amd/libopfx_dup$k.so', ['-DOPFX_DUP=$((1<<k))'], $units)"
  done
  ls opfgym_amd/libopfx_dup*.so | wc -l
else
  n=${2:-2}
  for i in $(seq $n); do
    echo "none: $(OPFX_LIB=opfgym_amd/libopfx_dupnone.so python scripts/probe_step.py 8192 2 2>&1 | grep k_step | tail -1)"
    for k in $(seq 0 11); do
      echo "${names[$k]}: $(OPFX_LIB=opfgym_amd/libopfx_dup$k.so python scripts/probe_step.py 8192 2 2>&1 | grep k_step | tail -1)"
    done
  done
fi
