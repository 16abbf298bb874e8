#!/bin/bash
# What-if for VERDICT r04 #4 (three teams of two per CU on the 306-bus grid): how the wave-team step kernel scales with the
# number of resident teams per CU, for teams of 2 and 4 (developer switches OPFX_TEAM, OPFX_WAVES_PER_CU; same box):
#   scripts/probe_team_occupancy.sh "<configs>"
for c in ${1:-3}; do
  st=10; [ $c = 5 ] && st=2
  for team in 4 2; do
    for per_cu in 1 2 3 4; do
      [ $team = 4 ] && [ $per_cu -gt 2 ] && continue
      OPFX_TEAM=$team OPFX_WAVES_PER_CU=$per_cu python bench.py --config $c --steps $st --warmup 2 --windows 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
lines=[l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')]
if not lines: print('config $c team $team per_cu $per_cu: no line (does not fit?)'); sys.exit()
d=json.loads(lines[-1]); kl=d['config']['kernel_launch']
print('config $c team $team requested per_cu $per_cu -> waves/instance', kl['waves_per_instance'], 'resident/CU', kl['instances_per_cu'], 'lds', kl['lds_bytes_per_instance'], 'kernel_ms %.4f' % d['roofline']['kernel_ms'], 'step/s %.0f' % d['value'])"
    done
  done
done
