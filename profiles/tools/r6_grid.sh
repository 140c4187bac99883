#!/bin/bash
# r06: cfg3 with fewer striding workgroups than segments (FW_SEG_GRID), one box
export FW_KNOBS=1
O=gpurun_out/r6_grid; mkdir -p $O; : > $O/ab.txt
run() { env "$@" timeout 200 python bench.py --config cfg3 --steps 8 --warmup 2 --no-cpu-baseline --no-other-schedule --no-one-chain 2>$O/err.txt | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('$*', round(l['ms_per_step'],2), l['edges'], l['network_sha256'][:12])" | tee -a $O/ab.txt; }
for i in 1 2; do
run FW_X=0
run FW_SEG_GRID=1024
run FW_SEG_GRID=2048
done
