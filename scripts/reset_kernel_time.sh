#!/bin/bash
# Duration of the reset kernel(s) under rocprofv3 for one BASELINE configuration (run on the GPU box through gpurun):
#   scripts/reset_kernel_time.sh <tag> [config]   ->  gpurun_out/reset_<tag>_kernel_stats.csv  (+ the same with OPFX_RESET_GLOBAL=1)
set -u
tag=${1:-latest}; cfg=${2:-2}
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/reset_$tag -- python3 $root/bench.py --config $cfg --steps 20 --no-cpu-baseline > $out/reset_$tag.log 2>&1
OPFX_RESET_GLOBAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/reset_${tag}_global -- python3 $root/bench.py --config $cfg --steps 20 --no-cpu-baseline > $out/reset_${tag}_global.log 2>&1
cd $root
for v in "" _global; do
  f=$(find $out/reset_$tag$v -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $out/reset_$tag${v}_kernel_stats.csv && grep -E "k_reset|k_step" $out/reset_$tag${v}_kernel_stats.csv | cut -c1-60,150-400
  grep '^{' $out/reset_$tag$v.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('cycle', c['reset_plus_step_ms'], 'step', d['ms_per_step'])"
done
