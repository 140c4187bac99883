#!/bin/bash
# NOT RUN YET (the round's GPU budget ended): cfg5 with striding workgroups in the max_k 4-5 segment kernels, the knob that gave cfg3 2 % (profiles/r06_cfg3_segment_grid.txt).
# The long-list kernel holds 3 workgroups per CU (768 resident), the level-2 table kernel 4 (1 024): candidates are 1 536 / 2 048 / 3 072 against the default (segments + 512 = 8 704).
# Two runs of one setting differ by up to 2 % at cfg5: run every setting twice.
export FW_KNOBS=1
O=gpurun_out/next_cfg5_grid; mkdir -p $O; : > $O/ab.txt
run() { env "$@" timeout 900 python bench.py --config cfg5 --steps 1 --warmup 0 --no-cpu-baseline --no-other-schedule --no-one-chain 2>$O/err.txt | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('cfg5 $*', round(l['ms_per_step'],1), l['edges'], l['network_sha256'][:12])" | tee -a $O/ab.txt; }
for i in 1 2; do
run FW_X=0
run FW_SEG_GRID=1536
run FW_SEG_GRID=2048
run FW_SEG_GRID=3072
done
