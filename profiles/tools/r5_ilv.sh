#!/bin/bash
# r05: interleaved lane <-> rank mapping of the size-3 table kernel only for chunks that start below a rank (FW_FZ_ILV_RANKS), runs of
# consecutive ranks per lane beyond; same box, alternating
export FW_KNOBS=1
O=gpurun_out/r5_ilv; mkdir -p $O; : > $O/ab.txt
for i in 1 2; do
for v in 18446744073709551615 0 2048 34816 200000 1000000; do
  FW_FZ_ILV_RANKS=$v timeout 300 python bench.py --config cfg3 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); r=l['roofline']; print('ilv_ranks $v', round(l['ms_per_step'],2), round(l['other_schedule']['ms_per_step'],2), l['edges'], '%.5g'%l['tests_per_step']['conditional_evaluated'], 'one-chain kernel s %.4f, evaluated/s in kernel %.4g'%(r['kernel_seconds_per_step'], r['evaluated_tests_per_s_in_kernel']))" | tee -a $O/ab.txt
done; done
