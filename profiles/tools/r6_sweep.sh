#!/bin/bash
# r06: the round schedule's knobs were swept in r03 / r04, when a cfg3 pass took 300 ms; the kernels are twice as fast now.  One box, cfg3 headline, every setting twice.
export FW_KNOBS=1
O=gpurun_out/r6_sweep; mkdir -p $O; : > $O/sweep.txt
run() { env "$@" timeout 300 python bench.py --config cfg3 --steps 8 --warmup 2 --no-cpu-baseline --no-other-schedule --no-one-chain 2>$O/err.txt | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('$*', round(l['ms_per_step'],2), l['edges'], '%.5g'%l['tests_per_step']['conditional_evaluated'], l['network_sha256'][:12])" | tee -a $O/sweep.txt; }
for i in 1 2; do
run FW_X=0
run FW_SEG_TARGET=2048
run FW_SEG_TARGET=4096
run FW_SEG_TARGET=6144
run FW_SEG_A=4000000 FW_SEG_B=8000000
run FW_SEG_A=12000000 FW_SEG_B=20000000
run FW_SEG_A=0 FW_SEG_B=0
run FW_DH_SPEC=2
run FW_DH_SPEC=6
run FW_DH_SPEC0=0
run FW_DH_SPEC0=4
run FW_W0_BIG=16384
run FW_W0_BIG=65536
run FW_DH_BATCH=2
run FW_DH_BATCH=8
done
