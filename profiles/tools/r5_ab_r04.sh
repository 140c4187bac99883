#!/bin/bash
# r05: the round-4 library (commit 1963beb, built from a worktree into flashweave.jl_amd/libfw_r04.so) against the current one on the SAME box:
# cfg3 headline / other schedule / one-chain kernel seconds, alternating
export FW_KNOBS=1
O=gpurun_out/r5_ab_r04; mkdir -p $O
for i in 1 2; do
for lib in libfw_r04.so libflashweave_amd.so; do
  FW_LIB_PATH=$PWD/flashweave.jl_amd/$lib timeout 300 python bench.py --config cfg3 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); r=l['roofline']; print('$lib', round(l['ms_per_step'],2), round(l['other_schedule']['ms_per_step'],2), l['edges'], '%.5g'%l['tests_per_step']['conditional_evaluated'], 'one-chain kernel s %.4f of step %.4f, evaluated/s in kernel %.4g'%(r['kernel_seconds_per_step'], r['step_seconds_of_that_pass'], r['evaluated_tests_per_s_in_kernel']))" | tee -a $O/ab.txt
done; done
