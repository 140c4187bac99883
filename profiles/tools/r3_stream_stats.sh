#!/bin/bash
# variant S inside learn_network (whole cfg3): kernel split
cd $GRAFT_REPO_ROOT; R=$PWD; export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/sstats
timeout 800 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sstats -- python $R/bench.py --stream-columns --steps 1 --warmup 0 --no-other-schedule --no-cpu-baseline > /tmp/sstats.json 2>/dev/null
find /tmp/sstats -name '*kernel_stats.csv' -exec head -6 {} \; | cut -c1-90,200-330
find /tmp/sstats -name '*kernel_stats.csv' -exec cp {} $R/gpurun_out/stream_whole_kernel_stats.csv \;
python -c "
import json; d=json.loads(open('/tmp/sstats.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['edges'], d['tests_per_step'], d['kernel_launches_per_step'])"
