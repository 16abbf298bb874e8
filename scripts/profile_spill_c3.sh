#!/bin/bash
# what the 208 B of scratch per lane of k_step<2,4,SPEC=2,MINW=3> (config 3, three teams of four per CU) cost in
# vector-memory instructions: the same shared-slot plan on the kernel compiled for TWO wavefronts per SIMD (no scratch; forced with
# OPFX_TEAM=4 OPFX_PLAN_SHARE=1: two teams per CU) against the default.  rocprofv3 --pmc, one pass per build.
#   scripts/profile_spill_c3.sh   (GPU box) -> gpurun_out/r06_spill_c3.txt
root=$(pwd); out=$root/gpurun_out/spill_c3; mkdir -p $out; export TMPDIR=/tmp; cd /tmp
pass() {  # label
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $out/$1 -- python3 $root/bench.py --config 3 --steps 5 --warmup 2 --windows 1 --no-cpu-baseline > $out/$1.log 2>&1
}
pass minw3
export OPFX_TEAM=4 OPFX_PLAN_SHARE=1
pass minw2
unset OPFX_TEAM OPFX_PLAN_SHARE
cd $root
python3 - $out <<'PY' | tee gpurun_out/r06_spill_c3.txt
import sys, glob, csv, collections, json
out = sys.argv[1]
print('# rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES, mean per k_step launch, bench.py --config 3 (8192 instances, 306 buses,')
print('# the plan with 208 shared LDS slots): the default kernel compiled for three wavefronts per SIMD (168 VGPRs + 208 B scratch per lane, three teams')
print('# of four per CU) against the one compiled for two (221 VGPRs, no scratch, two teams per CU; OPFX_TEAM=4 OPFX_PLAN_SHARE=1)')
for lab in ('minw3', 'minw2'):
    acc = collections.defaultdict(list); name = ''
    for f in glob.glob(f'{out}/{lab}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'k_step' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value'])); name = r['Kernel_Name']
    ms = None
    try:
        ms = json.loads([l for l in open(f'{out}/{lab}.log') if l.startswith('{')][-1])['ms_per_step']
    except Exception:
        pass
    print(lab, name.split('(')[0][-60:], {k: round(sum(v) / len(v)) for k, v in sorted(acc.items())}, 'ms_per_step (under the profiler)', ms)
PY
