#!/bin/bash
# Everything profiles/tools/install_profiles.py needs, in one gpurun call:  gpurun -- 'bash profiles/tools/collect_all.sh'
cd "${GRAFT_REPO_ROOT:-.}"
bash profiles/tools/collect_cfg3.sh > /dev/null 2>&1
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/final
timeout 900 python bench.py 2>/dev/null > gpurun_out/final/bench_cfg3.json
timeout 600 python bench.py --config cfg2 --steps 5 --warmup 1 2>/dev/null > gpurun_out/final/bench_cfg2.json
timeout 600 python bench.py --config cfg4 --steps 3 --warmup 1 2>/dev/null > gpurun_out/final/bench_cfg4.json
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof4 -- python $R/bench.py --config cfg4 --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/final/cfg4_bench_under_rocprof.json 2>/dev/null
find /tmp/prof4 -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/final/cfg4_kernel_stats.csv \;
ls -la $R/gpurun_out/final $R/gpurun_out/prof_cfg3
