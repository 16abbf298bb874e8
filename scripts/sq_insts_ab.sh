#!/bin/bash
# Instruction counters of bench.py's k_step for several builds on one GPU box (per launch: vector / scalar / LDS / branch
# instructions, kernel time): the per-flag deltas of the SPEC specialisations (builds with -DOPFX_SPEC_MASK=0|1|2|3).
#   scripts/sq_insts_ab.sh <config> libA.so libB.so ...   ->  gpurun_out/sq_insts_c<config>.txt
cfg=$1; shift
root=$(pwd); out=$root/gpurun_out/sq_insts_c$cfg.txt; : > $out
export TMPDIR=/tmp
cd /tmp
for lib in "$@"; do
  d=$root/gpurun_out/sq_insts_tmp; rm -rf $d
  OPFX_LIB=$root/$lib rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $d -- python3 $root/bench.py --config $cfg --steps 5 --warmup 2 --windows 1 --no-cpu-baseline > $d.log 2>&1
  t=$(OPFX_LIB=$root/$lib python3 $root/bench.py --config $cfg --steps 20 --warmup 3 --windows 3 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f %s' % (d['roofline']['kernel_ms'], d['roofline']['kernel']))")
  python3 - "$d" "$lib" "$t" >> $out <<'PY'
import sys, glob, csv, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_step' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
m = {k: sum(v) / len(v) for k, v in acc.items()}
print(sys.argv[2], 'kernel_ms', sys.argv[3], ' '.join('%s=%.2fM' % (k.replace('SQ_INSTS_', ''), v / 1e6) for k, v in sorted(m.items())))
PY
done
cat $out
