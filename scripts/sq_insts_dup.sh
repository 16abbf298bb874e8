#!/bin/bash
# Instruction counts of each prologue / epilogue phase of k_step: the -DOPFX_DUP builds of scripts/ab_dup.sh under the SQ
# instruction counters; (count with phase k doubled) - (count of the plain build) = instructions of phase k per launch.
#   scripts/ab_dup.sh build; then on the GPU box: scripts/sq_insts_dup.sh   ->  gpurun_out/sq_insts_dup.txt
names=(stage_row actions table_obs injections cost_pre init_voltage compute_results derived_rows constraints cost_post results_out obs_out)
root=$(pwd); out=$root/gpurun_out/sq_insts_dup.txt; : > $out
export TMPDIR=/tmp
cd /tmp
for k in none $(seq 0 11); do
  d=$root/gpurun_out/sq_dup_tmp; rm -rf $d
  OPFX_LIB=$root/opfgym_amd/libopfx_dup$k.so rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $d -- python3 $root/scripts/probe_step.py 8192 2 > $d.log 2>&1
  name=$k; [ "$k" != none ] && name=${names[$k]}
  python3 - "$d" "$name" >> $out <<'PY'
import sys, glob, csv, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_step' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
m = {k: sum(v) / len(v) / 8192 for k, v in acc.items()}
print('%-16s' % sys.argv[2], ' '.join('%s=%.0f' % (k.replace('SQ_INSTS_', ''), v) for k, v in sorted(m.items())), '(per instance)')
PY
done
rm -rf $root/gpurun_out/sq_dup_tmp $root/gpurun_out/sq_dup_tmp.log
cat $out
