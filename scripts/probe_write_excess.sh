#!/bin/bash
# Where the HBM write excess of a config-2 launch comes from (EXPERIMENTS #33): WRITE_SIZE of the product library against a probe
# build that leaves the scattered 8-byte set-point stores into x out (-DOPFX_PROBE_NO_SETPOINT_WRITEBACK; results then wrong on
# purpose).  scripts/probe_write_excess.sh (GPU box) -> gpurun_out/r06_write_excess.txt
root=$(pwd); out=$root/gpurun_out/write_excess; mkdir -p $out; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/full -- python3 $root/bench.py --steps 5 --warmup 2 --windows 1 --no-cpu-baseline --no-also > $out/full.log 2>&1
export OPFX_LIB=$root/opfgym_amd/libopfx_nowb.so
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/nowb -- python3 $root/bench.py --steps 5 --warmup 2 --windows 1 --no-cpu-baseline --no-also > $out/nowb.log 2>&1
unset OPFX_LIB; cd $root
python3 - $out <<'PY' | tee gpurun_out/r06_write_excess.txt
import sys, glob, csv
for lab in ('full', 'nowb'):
    v = [float(r['Counter_Value']) for f in glob.glob(f'{sys.argv[1]}/{lab}/**/*counter_collection.csv', recursive=True) for r in csv.DictReader(open(f)) if 'k_step' in r['Kernel_Name'] and r['Counter_Name'] == 'WRITE_SIZE']
    print(lab, 'WRITE_SIZE per k_step launch: %.1f MB (n=%d)' % (sum(v) / len(v) * 1024 / 1e6, len(v)))
PY
