#!/bin/bash
# r05: the previous build (flashweave.jl_amd/libfw_prev.so, built from HEAD before the change under test) against the current one on the SAME box
export FW_KNOBS=1 FW_TRACE_HOST=1
O=gpurun_out/r5_ab_prev; mkdir -p $O; : > $O/ab.txt
for cfg in cfg3 cfg4; do
for i in 1 2; do
for lib in libfw_prev.so libflashweave_amd.so; do
  FW_LIB_PATH=$PWD/flashweave.jl_amd/$lib timeout 400 python bench.py --config $cfg --steps 8 --warmup 2 --no-cpu-baseline 2>$O/err.txt | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('$cfg $lib', round(l['ms_per_step'],2), (l.get('other_schedule') or {}).get('ms_per_step'), l['edges'])" | tee -a $O/ab.txt
  grep "symmetric graph" $O/err.txt | tail -2 | tee -a $O/ab.txt
done; done; done
timeout 900 python -m pytest tests/test_gpu_fz.py tests/test_gpu_fuzz.py tests/test_gpu_fznz.py -m gpu -q -x 2>&1 | tail -3 | tee -a $O/ab.txt
