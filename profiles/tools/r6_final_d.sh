#!/bin/bash
# r06 final collection D (after the cfg5 changes: local matrices for max_k 4-5, 8 192 segments per launch): cfg5's bench line and kernel trace, then the whole GPU suite
mkdir -p gpurun_out/r6_final_d
timeout 900 python bench.py --config cfg5 --steps 1 --warmup 0 --no-cpu-baseline 2>gpurun_out/r6_final_d/err.txt | tail -1 > gpurun_out/r6_final_d/bench_cfg5.json
python -c "import json; l=json.load(open('gpurun_out/r6_final_d/bench_cfg5.json')); print('cfg5', l['ms_per_step'], l['edges'], l['network_sha256'][:12])"
R=$PWD; P=$R/gpurun_out/r6_final_d
( cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof5; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof5 -- python $R/bench.py --config cfg5 --no-cpu-baseline --no-other-schedule --no-one-chain --steps 1 --warmup 0 > $P/bench_under_rocprof.json 2>/tmp/prof5.err; find /tmp/prof5 -name '*kernel_stats.csv' -exec cp {} $P/kernel_stats.csv \; )
( time timeout 1100 python -m pytest tests -q -m gpu --durations=15 ) > gpurun_out/r6_final_d/pytest.txt 2>&1; tail -25 gpurun_out/r6_final_d/pytest.txt
