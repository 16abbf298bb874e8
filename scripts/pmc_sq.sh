#!/bin/bash
# SQ counter passes over bench.py's k_step launches (run on the GPU box through gpurun).
# Usage: scripts/pmc_sq.sh <tag>       -> gpurun_out/pmc_<tag>/summary.txt
set -u
tag=${1:-sq}
root=$(pwd)
out=$root/gpurun_out/pmc_$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
           "SQ_WAVES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- python3 $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/p$i.log 2>&1
done
cd $root
python3 - "$out" <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Kernel_Name'].startswith('k_step') or 'k_step' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
with open(out + '/summary.txt', 'w') as fh:
    for k in sorted(acc):
        v = acc[k]
        fh.write('%-28s %16.0f  (n=%d)\n' % (k, sum(v) / len(v), len(v)))
print(open(out + '/summary.txt').read())
PY
