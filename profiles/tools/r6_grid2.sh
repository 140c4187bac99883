#!/bin/bash
# r06: cfg3, striding workgroups of the segment kernel (FW_SEG_GRID; default = segments + 512 = 3 584), one box
export FW_KNOBS=1
O=gpurun_out/r6_grid2; mkdir -p $O; : > $O/ab.txt
run() { env "$@" timeout 200 python bench.py --config cfg3 --steps 8 --warmup 2 --no-cpu-baseline --no-one-chain 2>$O/err.txt | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('$*', round(l['ms_per_step'],2), 'ff=0', round(l['other_schedule']['ms_per_step'],2), l['edges'], l['network_sha256'][:12])" | tee -a $O/ab.txt; }
run FW_SEG_GRID=1536
run FW_SEG_GRID=1792
run FW_SEG_GRID=2048
run FW_SEG_GRID=2304
run FW_SEG_GRID=2560
run FW_X=0
