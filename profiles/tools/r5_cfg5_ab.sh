#!/bin/bash
# r05: cfg5 (max_k 5, long lists) -- NaN-free arithmetic decided per test instead of per chunk -- against the previous build on one box, then the parity tests of that path
export FW_KNOBS=1
O=gpurun_out/r5_cfg5_ab; mkdir -p $O; : > $O/ab.txt
timeout 900 python -m pytest tests/test_gpu_fz.py tests/test_gpu_fzs.py -m gpu -q -x 2>&1 | tail -2 | tee -a $O/ab.txt
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k "cfg5" 2>&1 | tail -2 | tee -a $O/ab.txt
for lib in libfw_prev.so libflashweave_amd.so; do
  FW_LIB_PATH=$PWD/flashweave.jl_amd/$lib timeout 600 python bench.py --config cfg5 --steps 1 --warmup 0 --no-cpu-baseline --no-other-schedule --no-one-chain 2>/dev/null | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('$lib', 'cfg5 s %.2f'%(l['ms_per_step']/1e3), l['edges'], l['tests_per_step'])" | tee -a $O/ab.txt
done
