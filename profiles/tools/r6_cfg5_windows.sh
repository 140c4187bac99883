#!/bin/bash
# r06: cfg5's first windows (FW_W0_SMALL: jobs with fewer than 64 accepted variables, 256 for max_k > 3; FW_W0_BIG: the others, 16 384) -- never measured at full cfg5 size.  One box, one pass each.
export FW_KNOBS=1
O=gpurun_out/r6_cfg5_windows; mkdir -p $O; : > $O/ab.txt
run() { env "$@" timeout 900 python bench.py --config cfg5 --steps 1 --warmup 0 --no-cpu-baseline --no-other-schedule --no-one-chain 2>$O/err.txt | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('cfg5 $*', round(l['ms_per_step'],1), l['edges'], l['network_sha256'][:12], 'evaluated %.5g'%l['tests_per_step']['conditional_evaluated'], 'launches', l['kernel_launches_per_step'])" | tee -a $O/ab.txt; }
run FW_X=0
run FW_W0_SMALL=2048 FW_W0_BIG=65536
run FW_W0_SMALL=2048
run FW_W0_BIG=65536
