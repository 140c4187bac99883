#!/bin/bash
# r06: A/B of the current build against libfw_row0.so (the previous build), cfg3, one box; then the parity suites
export FW_KNOBS=1
O=gpurun_out/r6_row; mkdir -p $O; : > $O/ab.txt
run() { lib=$1; shift; env "$@" FW_LIB_PATH=$PWD/flashweave.jl_amd/$lib timeout 400 python bench.py --config cfg3 --steps 6 --warmup 2 --no-cpu-baseline 2>$O/err.txt | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); r=l['roofline']; print('cfg3 $lib $*', round(l['ms_per_step'],2), round((l.get('other_schedule') or {}).get('ms_per_step',0),2), l['edges'], '%.5g'%l['tests_per_step']['conditional_evaluated'], 'kernel s %.4f (%s), evaluated/s in kernel %.4g'%(r['kernel_seconds_per_step'], r['measured_on'][:9], r['evaluated_tests_per_s_in_kernel']), l['network_sha256'][:12])" | tee -a $O/ab.txt; tail -3 $O/err.txt | grep -i "error\|fail" ; }
for i in 1 2; do
run libfw_row0.so FW_X=0
run libflashweave_amd.so FW_X=0
done
timeout 1200 python -m pytest tests/test_gpu_fz.py tests/test_gpu_fznz.py tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -3 | tee -a $O/ab.txt
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k "cfg3" 2>&1 | tail -3 | tee -a $O/ab.txt
