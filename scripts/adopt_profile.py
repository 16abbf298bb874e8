"""Copy the summaries of one `scripts/profile_round.sh <tag> <config>` run from gpurun_out/ (scratch) into profiles/
(tracked) and make them the entry bench.py reads for that configuration (profiles/pmc_latest.json).

    python scripts/adopt_profile.py <tag> [<tag> ...]
"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
latest_path = os.path.join(ROOT, 'profiles', 'pmc_latest.json')
latest = json.load(open(latest_path)) if os.path.exists(latest_path) else {}
for tag in sys.argv[1:]:
    src = os.path.join(ROOT, 'gpurun_out', f'prof_{tag}')
    for name, dst in (('kernel_stats.csv', f'{tag}_bench_kernel_stats.csv'), ('pmc_summary.json', f'{tag}_pmc_summary.json'),
                      ('sq_counters.txt', f'{tag}_sq_counters.txt')):
        if os.path.exists(os.path.join(src, name)):
            shutil.copy(os.path.join(src, name), os.path.join(ROOT, 'profiles', dst))
    summary = json.load(open(os.path.join(src, 'pmc_summary.json')))
    latest[f'config{summary["config"]}'] = summary
    print(tag, '-> config', summary['config'], 'batch', summary.get('batch'), 'kernel avg ns', summary.get('kernel_avg_ns'))
json.dump(latest, open(latest_path, 'w'), indent=1)
