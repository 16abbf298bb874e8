#!/bin/bash
# Instruction counters of k_step at FIXED Newton iteration counts (tol = 0, max_iter = n): the difference of two counts is
# the cost of one iteration, the count at 0 is prologue + one mismatch pass + epilogue.
#   scripts/sq_insts_fixed_it.sh <config> <batch> it0 it1 ...   ->  gpurun_out/sq_insts_fixed_c<config>.txt
cfg=$1; B=$2; shift 2
root=$(pwd); out=$root/gpurun_out/sq_insts_fixed_c$cfg.txt; : > $out
export TMPDIR=/tmp
cd /tmp
for it in "$@"; do
  d=$root/gpurun_out/sq_fixed_tmp; rm -rf $d
  OPFX_FIXED_IT=$it rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $d -- python3 $root/scripts/probe_step.py $B $cfg > $d.log 2>&1
  python3 - "$d" "$it" "$(tail -1 $d.log)" >> $out <<'PY'
import sys, glob, csv, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_step' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
m = {k: sum(v) / len(v) for k, v in acc.items()}
print('it', sys.argv[2], '|', sys.argv[3], '|', ' '.join('%s=%.3fM' % (k.replace('SQ_INSTS_', ''), v / 1e6) for k, v in sorted(m.items())))
PY
done
cat $out
