#!/bin/bash
# r05 run J: kernel trace of cfg4 (level-0 stage composition after the matrix-loop work)
O=$PWD/gpurun_out/r5_j; mkdir -p $O
R=$PWD
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_stats
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python $R/bench.py --config cfg4 --no-cpu-baseline --no-other-schedule --no-one-chain --steps 3 --warmup 1 > $O/bench_under_rocprof.json 2> /tmp/prof_stats.err
find /tmp/prof_stats -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats.csv \;
head -25 $O/kernel_stats.csv | cut -c1-200
