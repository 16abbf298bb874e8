#!/bin/bash
# SQ wait/issue counters of bench.py's k_step for several builds on one GPU box:
#   scripts/sq_ab.sh <config> libA.so libB.so ...   ->  gpurun_out/sq_ab_c<config>.txt
cfg=$1; shift
root=$(pwd); out=$root/gpurun_out/sq_ab_c$cfg.txt; : > $out
export TMPDIR=/tmp
cd /tmp
for lib in "$@"; do
  d=$root/gpurun_out/sq_ab_tmp; rm -rf $d
  OPFX_LIB=$root/$lib rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $d -- python3 $root/bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline > $d.log 2>&1
  python3 - "$d" "$lib" >> $out <<'PY'
import sys, glob, csv, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_step' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
m = {k: sum(v) / len(v) for k, v in acc.items()}
print(sys.argv[2], ' '.join('%s=%.0f' % kv for kv in sorted(m.items())))
if 'SQ_WAVE_CYCLES' in m:
    print('   wait/wave_cycles %.3f   active/wave_cycles %.3f   valu/wave_cycles %.3f' % (
        m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES'], m['SQ_ACTIVE_INST_ANY'] / m['SQ_WAVE_CYCLES'], m['SQ_ACTIVE_INST_VALU'] / m['SQ_WAVE_CYCLES']))
PY
done
cat $out
