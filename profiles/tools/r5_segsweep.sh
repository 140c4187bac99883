#!/bin/bash
# r05: segments per launch (FW_SEG_TARGET) and the launch-size steps (FW_SEG_A / FW_SEG_B) on the kernel with the fast loop -- the per-segment
# fixed cost (table build, unranking, reductions) is now ~39 % of the kernel's vector instructions
export FW_KNOBS=1
O=gpurun_out/r5_segsweep; mkdir -p $O; : > $O/sweep.txt
run() { env "$@" timeout 300 python bench.py --config cfg3 --steps 5 --warmup 2 --no-cpu-baseline --no-one-chain 2>/dev/null | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('$*', round(l['ms_per_step'],2), round(l['other_schedule']['ms_per_step'],2), l['edges'], '%.5g'%l['tests_per_step']['conditional_evaluated'])" | tee -a $O/sweep.txt; }
run FW_X=0
run FW_SEG_TARGET=2048
run FW_SEG_TARGET=1536
run FW_SEG_TARGET=1024
run FW_SEG_TARGET=4096
run FW_SEG_A=16000000 FW_SEG_B=24000000
run FW_SEG_A=32000000 FW_SEG_B=48000000
run FW_SEG_TARGET=2048 FW_SEG_A=16000000 FW_SEG_B=24000000
run FW_X=0
