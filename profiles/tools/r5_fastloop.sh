#!/bin/bash
# r05: the common size-3 test behind one wave-uniform branch (FW_FZ_FASTLOOP) against the previous build (flashweave.jl_amd/libfw_prev.so) on one box
export FW_KNOBS=1
O=gpurun_out/r5_fastloop; mkdir -p $O; : > $O/ab.txt
for i in 1 2; do
for lib in libfw_prev.so libflashweave_amd.so; do
  FW_LIB_PATH=$PWD/flashweave.jl_amd/$lib timeout 300 python bench.py --config cfg3 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); r=l['roofline']; print('$lib', round(l['ms_per_step'],2), round(l['other_schedule']['ms_per_step'],2), l['edges'], '%.5g'%l['tests_per_step']['conditional_evaluated'], 'one-chain kernel s %.4f, evaluated/s in kernel %.4g'%(r['kernel_seconds_per_step'], r['evaluated_tests_per_s_in_kernel']))" | tee -a $O/ab.txt
done; done
timeout 1200 python -m pytest tests/test_gpu_fz.py tests/test_gpu_fuzz.py tests/test_gpu_fznz.py -m gpu -q -x 2>&1 | tail -3 | tee -a $O/ab.txt
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k "cfg3" 2>&1 | tail -3 | tee -a $O/ab.txt
