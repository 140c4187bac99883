#!/bin/bash
# r06: cfg5 with local matrices, variants on one box: the level-2 table kernel (short lists) without them (hk2), the size-5 gather with a 32-bit row base (row), both; smallest degree 1
export FW_KNOBS=1
O=gpurun_out/r6_cfg5_tmat2; mkdir -p $O; : > $O/ab.txt
run() { lib=$1; shift; env "$@" FW_LIB_PATH=$PWD/flashweave.jl_amd/$lib timeout 900 python bench.py --config cfg5 --steps 1 --warmup 0 --no-cpu-baseline --no-other-schedule --no-one-chain 2>$O/err.txt | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('cfg5 $lib $*', round(l['ms_per_step'],1), l['edges'], l['network_sha256'][:12], 'kernel s', round(l['roofline']['kernel_seconds_per_step'],2))" | tee -a $O/ab.txt; }
run libflashweave_amd.so FW_X=0
run libfw_v_hk2.so FW_X=0
run libfw_v_row.so FW_X=0
run libfw_v_hk2row.so FW_X=0
run libflashweave_amd.so FW_FZ_TMAT=1
