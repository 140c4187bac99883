#!/bin/bash
# r06: cfg5, more segments per launch (FW_SEG_TARGET; 3 072 was cfg3's optimum), one box
export FW_KNOBS=1
O=gpurun_out/r6_cfg5_knobs2; mkdir -p $O; : > $O/ab.txt
run() { env "$@" timeout 900 python bench.py --config cfg5 --steps 1 --warmup 0 --no-cpu-baseline --no-other-schedule --no-one-chain 2>$O/err.txt | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('cfg5 $*', round(l['ms_per_step'],1), l['edges'], l['network_sha256'][:12], 'kernel s', round(l['roofline']['kernel_seconds_per_step'],2))" | tee -a $O/ab.txt; }
run FW_SEG_TARGET=6144
run FW_SEG_TARGET=8192
run FW_SEG_TARGET=12288
run FW_SEG_TARGET=16384
run FW_X=0
