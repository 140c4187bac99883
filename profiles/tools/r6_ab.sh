#!/bin/bash
# r06: A/B of library builds on ONE box: r6_ab.sh <config> <reps> <lib> <lib> ...   (libs under flashweave.jl_amd/)
export FW_KNOBS=1
cfg=$1; reps=$2; shift 2
O=gpurun_out/r6_ab; mkdir -p $O
for i in $(seq 1 $reps); do
for lib in "$@"; do
  FW_LIB_PATH=$PWD/flashweave.jl_amd/$lib timeout 400 python bench.py --config $cfg --steps 6 --warmup 2 --no-cpu-baseline 2>$O/err.txt | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); r=l['roofline']; print('$cfg $lib', round(l['ms_per_step'],2), round((l.get('other_schedule') or {}).get('ms_per_step',0),2), l['edges'], '%.5g'%l['tests_per_step']['conditional_evaluated'], 'kernel s %.4f (%s), evaluated/s in kernel %.4g'%(r['kernel_seconds_per_step'], r['measured_on'][:9], r['evaluated_tests_per_s_in_kernel']))" | tee -a $O/ab_$cfg.txt
done; done
