#!/bin/bash
# k_reset's own duration (rocprofv3 kernel trace) for the config-2 environment:  scripts/reset_kernel_time.sh [lib.so]
root=$(pwd); export TMPDIR=/tmp; d=$root/gpurun_out/reset_trace; rm -rf $d
cd /tmp
OPFX_LIB=${1:+$root/$1} rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $root/scripts/probe_reset.py 8192 > /dev/null 2>&1
python3 - "$d" <<'PY'
import sys, glob, csv
for f in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_reset' in r['Name'] or 'k_step' in r['Name']:
            print(r['Name'][:40], r['Calls'], 'avg %.1f us' % (float(r['AverageNs']) / 1e3))
PY
