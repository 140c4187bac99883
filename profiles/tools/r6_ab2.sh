#!/bin/bash
# r06: current build against libfw_head.so (the committed build), cfg3, one box
export FW_KNOBS=1
O=gpurun_out/r6_ab2; mkdir -p $O; : > $O/ab.txt
run() { lib=$1; shift; env "$@" FW_LIB_PATH=$PWD/flashweave.jl_amd/$lib timeout 400 python bench.py --config cfg3 --steps 6 --warmup 2 --no-cpu-baseline 2>$O/err.txt | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); r=l['roofline']; print('cfg3 $lib $*', round(l['ms_per_step'],2), round((l.get('other_schedule') or {}).get('ms_per_step',0),2), l['edges'], '%.5g'%l['tests_per_step']['conditional_evaluated'], 'kernel s %.4f (%s), evaluated/s in kernel %.4g'%(r['kernel_seconds_per_step'], r['measured_on'][:9], r['evaluated_tests_per_s_in_kernel']), l['network_sha256'][:12])" | tee -a $O/ab.txt; }
for i in 1 2; do
run libfw_head.so FW_X=0
run libfw_al64.so FW_X=0
done
