#!/bin/bash
# per-kernel averages of one bench run under rocprofv3:  bash profiles/tools/kstats.sh <tag> [bench args...]   (env passes through)
R=${GRAFT_REPO_ROOT:-$PWD}; tag=$1; shift
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ks_$tag
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$tag -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > /tmp/ks_$tag.json 2>/dev/null
python - <<PY
import csv,glob,json
f=glob.glob('/tmp/ks_$tag/**/*kernel_stats.csv',recursive=True)[0]
print('$tag', 'ms_per_step', json.loads(open('/tmp/ks_$tag.json').read().strip().splitlines()[-1])['ms_per_step'])
for r in csv.DictReader(open(f)):
    if float(r['TotalDurationNs'])>3e6: print('   %-50s calls %6s avg %8.1f us total %8.1f ms' % (r['Name'][:50], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
