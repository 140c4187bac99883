#!/bin/bash
# r06: A/B on one box: environment settings of the current build
export FW_KNOBS=1
O=gpurun_out/r6_ab2; mkdir -p $O; : > $O/ab.txt
run() { cfg=$1; shift; env "$@" timeout 400 python bench.py --config $cfg --steps 8 --warmup 2 --no-cpu-baseline 2>$O/err.txt | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); r=l['roofline']; print('$cfg $*', round(l['ms_per_step'],2), round((l.get('other_schedule') or {}).get('ms_per_step',0),2), l['edges'], '%.5g'%l['tests_per_step']['conditional_evaluated'], 'kernel s %.4f (%s)'%(r['kernel_seconds_per_step'], r['measured_on'][:9]), l['network_sha256'][:12])" | tee -a $O/ab.txt; }
for i in 1 2; do
run cfg3 FW_X=0
run cfg3 FW_DH_CHAINS=3
run cfg3he FW_DH_PLAN_SMALL=0
run cfg3he FW_X=0
done
