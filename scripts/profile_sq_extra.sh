#!/bin/bash
# Extra SQ/SQC counter passes for k_step (instruction cache, FP64 mix, LDS queue levels):
#   scripts/profile_sq_extra.sh <tag> [config]  ->  gpurun_out/prof_<tag>/sq_extra.txt
set -u
tag=${1:-latest}
cfg=${2:-2}
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_THREAD_CYCLES_VALU" \
           "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_LDS_ADDR_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL" \
           "SQ_ACTIVE_INST_VALU2 SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_LEVEL_WAVES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SMEM SQ_INST_LEVEL_SMEM SQC_DCACHE_MISSES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/sqx$i -- python3 $root/bench.py --config $cfg --steps 5 --warmup 2 --windows 1 --no-cpu-baseline > $out/sqx$i.log 2>&1
done
cd $root
python3 - "$out" "$cfg" <<'PY'
import sys, glob, csv, collections
out, cfg = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(out + '/sqx*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_step' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
with open(out + '/sq_extra.txt', 'w') as fh:
    fh.write('# rocprofv3 --pmc, mean per k_step launch, bench.py --config %s --steps 5 --warmup 2 --windows 1\n' % cfg)
    for k in sorted(acc):
        fh.write('%-30s %16.0f  (n=%d)\n' % (k, sum(acc[k]) / len(acc[k]), len(acc[k])))
print(open(out + '/sq_extra.txt').read())
PY
