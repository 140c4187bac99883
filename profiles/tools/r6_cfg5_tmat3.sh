#!/bin/bash
# r06: cfg5, new defaults (32-bit row base, matrices from degree 1) against an XCD-aware segment order in the max_k 4-5 kernels (xcd5), one box; parity tests of the max_k 4-5 paths first
export FW_KNOBS=1
O=gpurun_out/r6_cfg5_tmat3; mkdir -p $O; : > $O/ab.txt
timeout 900 python -m pytest tests/test_gpu_fz.py tests/test_gpu_maxk.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest.txt
run() { lib=$1; shift; env "$@" FW_LIB_PATH=$PWD/flashweave.jl_amd/$lib timeout 900 python bench.py --config cfg5 --steps 1 --warmup 0 --no-cpu-baseline --no-other-schedule --no-one-chain 2>$O/err.txt | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('cfg5 $lib $*', round(l['ms_per_step'],1), l['edges'], l['network_sha256'][:12], 'kernel s', round(l['roofline']['kernel_seconds_per_step'],2))" | tee -a $O/ab.txt; }
run libflashweave_amd.so FW_X=0
run libfw_v_xcd5.so FW_X=0
run libflashweave_amd.so FW_X=0
