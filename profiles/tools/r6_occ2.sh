#!/bin/bash
# r06: dh_mi_target_kernel compiled for two workgroups per CU (-DDH_MI_OCC=2: 248 VGPRs, no AGPRs, +130 B scratch) and launched with two per CU, against the default build, one box
export FW_KNOBS=1
O=gpurun_out/r6_occ2; mkdir -p $O; : > $O/ab.txt
run() { lib=$1; shift; for cfg in cfg4 cfg2; do env "$@" FW_LIB_PATH=$PWD/flashweave.jl_amd/$lib timeout 400 python bench.py --config $cfg --steps 8 --warmup 2 --no-cpu-baseline 2>$O/err.txt | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('$cfg $lib $*', round(l['ms_per_step'],2), round((l.get('other_schedule') or {}).get('ms_per_step',0),2), l['edges'], l['tests_per_step']['conditional_evaluated'], 'cond stage s', round(l['stage_seconds_rank0']['conditional'],4))" | tee -a $O/ab.txt; done; }
for i in 1 2; do
run libfw_miprev.so FW_X=0
run libflashweave_amd.so FW_X=0
run libfw_occ2.so FW_MI_WG_PER_CU=2
run libfw_occ2.so FW_MI_WG_PER_CU=1
done
