#!/bin/bash
# Instruction / scalar cache counters of the reset kernel (rocprofv3 --pmc) on one BASELINE configuration:
#   scripts/profile_reset_icache.sh <tag> [config]  ->  gpurun_out/reset_icache_<tag>.txt
set -u
tag=${1:-latest}; cfg=${2:-2}
root=$(pwd); out=$root/gpurun_out/reset_icache_$tag; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH" \
           "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_BRANCH" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- python3 $root/bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline > $out/p$i.log 2>&1
done
cd $root
python3 - "$out" <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_reset' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
with open(out + '.txt', 'w') as fh:
    fh.write('# rocprofv3 --pmc, mean per k_reset launch, bench.py --steps 5 --warmup 2\n')
    for k in sorted(acc):
        fh.write('%-26s %16.0f  (n=%d)\n' % (k, sum(acc[k]) / len(acc[k]), len(acc[k])))
print(open(out + '.txt').read())
PY
