#!/bin/bash
# r06: cfg5 with local matrices for the max_k 4-5 kernels: the bench line, the kernel trace and the counter passes
mkdir -p gpurun_out/r6_cfg5_collect
timeout 900 python bench.py --config cfg5 --steps 1 --warmup 0 --no-cpu-baseline 2>gpurun_out/r6_cfg5_collect/err.txt | tail -1 > gpurun_out/r6_cfg5_collect/bench_cfg5.json
python -c "import json; l=json.load(open('gpurun_out/r6_cfg5_collect/bench_cfg5.json')); print('cfg5', l['ms_per_step'], l['edges'], l['network_sha256'][:12])"
ROUND=r06 STATS_STEPS=1 STATS_WARMUP=0 PROF_TIMEOUT=1200 bash profiles/tools/collect_profile.sh cfg5 > gpurun_out/collect_cfg5.log 2>&1
ls gpurun_out/prof_r06_cfg5
