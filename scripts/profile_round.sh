#!/bin/bash
# Collects the rocprofv3 evidence for bench.py's k_step on the GPU box (run through gpurun):
#   scripts/profile_round.sh <tag> [config] [steps]
#     ->  gpurun_out/prof_<tag>/{kernel_stats.csv, pmc_summary.json, sq_counters.txt}
# Passes (never combined, MI355X_MICROARCH.md "rocprofv3 PMC slots"): kernel trace + stats;
# --pmc FETCH_SIZE; --pmc WRITE_SIZE; three SQ counter sets.
set -u
tag=${1:-latest}
cfg=${2:-2}
steps=${3:-20}
psteps=$(( steps < 5 ? steps : 5 ))
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/bench.py --config $cfg --steps $steps --windows 3 --no-cpu-baseline > $out/trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$c -- python3 $root/bench.py --config $cfg --steps $psteps --warmup 2 --windows 1 --no-cpu-baseline > $out/$c.log 2>&1
done
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
           "SQ_WAVES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/sq$i -- python3 $root/bench.py --config $cfg --steps $psteps --warmup 2 --windows 1 --no-cpu-baseline > $out/sq$i.log 2>&1
done
cd $root
python3 - "$out" "$tag" "$cfg" "$psteps" <<'PY'
import sys, glob, csv, collections, json, shutil, hashlib
out, tag, cfg, psteps = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
sha = hashlib.sha256()
for rel in ('opfgym_amd/csrc/opfx_dev.h', 'opfgym_amd/csrc/opfx.hip', 'opfgym_amd/csrc/plan.cpp', 'opfgym_amd/csrc/plan.h'):      # (kernels, host side, plan compiler)
    sha.update(open(rel, 'rb').read())          # (= bench.py source_sha16: counters are only used with the sources they were taken from)
st = glob.glob(out + '/trace/**/*kernel_stats.csv', recursive=True)
if st:
    shutil.copy(st[0], out + '/kernel_stats.csv')
acc = collections.defaultdict(list)
name = ''
for f in glob.glob(out + '/*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_step' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
            name = r['Kernel_Name']
mean = {k: sum(v) / len(v) for k, v in acc.items()}
with open(out + '/sq_counters.txt', 'w') as fh:
    fh.write('# rocprofv3 --pmc, mean per k_step launch, bench.py --config %s --steps %s --warmup 2\n# %s\n' % (cfg, psteps, name))
    for k in sorted(mean):
        if k not in ('FETCH_SIZE', 'WRITE_SIZE'):
            fh.write('%-28s %16.0f  (n=%d)\n' % (k, mean[k], len(acc[k])))
avg_ns, calls = None, None
if st:
    for r in csv.DictReader(open(st[0])):
        if 'k_step' in r.get('Name', ''):
            avg_ns, calls = float(r['AverageNs']), int(r['Calls'])
# min / median of the k_step launches of the stats pass (the kept profile's average has been above the
# driver's clock two rounds running; the spread says whether that is noise)
dur = []
for f in glob.glob(out + '/trace/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_step' in r.get('Kernel_Name', ''):
            dur.append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
dur.sort()
min_ns = dur[0] if dur else None
med_ns = dur[len(dur) // 2] if dur else None
batch = None
try:
    line = [ln for ln in open(out + '/trace.log') if ln.startswith('{')][-1]
    batch = json.loads(line)['config']['batch_per_gpu']
except Exception:
    pass
if 'FETCH_SIZE' in mean and 'WRITE_SIZE' in mean:
    fk, wk = mean['FETCH_SIZE'], mean['WRITE_SIZE']
    json.dump({'kernel': name, 'tag': tag, 'config': int(cfg), 'batch': batch, 'source_sha16': sha.hexdigest()[:16],
               'kernel_avg_ns': avg_ns, 'kernel_min_ns': min_ns, 'kernel_median_ns': med_ns, 'kernel_calls': calls,
               'sq': {k: v for k, v in mean.items() if k not in ('FETCH_SIZE', 'WRITE_SIZE')},
               'command': 'rocprofv3 --kernel-trace --pmc <FETCH_SIZE|WRITE_SIZE> --output-format csv -- python3 bench.py --config %s --steps %s --warmup 2 --no-cpu-baseline (one pass per counter)' % (cfg, psteps),
               'FETCH_SIZE_KB_per_launch': fk, 'WRITE_SIZE_KB_per_launch': wk,
               'FETCH_SIZE_launches': len(acc['FETCH_SIZE']), 'WRITE_SIZE_launches': len(acc['WRITE_SIZE']),
               'hbm_bytes_per_launch_raw': (fk + wk) * 1024, 'hbm_bytes_per_launch_fetch_x2': (2 * fk + wk) * 1024,
               'note': 'gfx950: FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM section); our reads are 8 B/lane, an uncalibrated width, so the true value lies between raw and fetch_x2 (bench.py reports fetch_x2). The round-2 kernels use no scratch memory.'},
              open(out + '/pmc_summary.json', 'w'), indent=1)
print(open(out + '/sq_counters.txt').read())
print(open(out + '/kernel_stats.csv').read()[:1500] if st else 'no kernel stats')
PY
