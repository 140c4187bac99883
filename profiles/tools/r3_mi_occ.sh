#!/bin/bash
# discrete persistent kernel: two workgroups per CU (VGPR + AGPR <= 256) against one
cd $GRAFT_REPO_ROOT
export FW_KNOBS=1
run() { name=$1; shift; cfg=$1; shift; ff=$1; shift
  env "$@" timeout 300 python bench.py --config $cfg --feed-forward $ff --steps 3 --warmup 1 --no-cpu-baseline --no-other-schedule --no-one-chain 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$name', round(d['ms_per_step'],2), 'cond', round(1e3*d['stage_seconds_rank0']['conditional'],2), 'eval', d['tests_per_step']['conditional_evaluated'], 'edges', d['edges'])"
}
for occ in 1 2; do
  touch flashweave.jl_amd/csrc/fw_devhiton.hip; make -C flashweave.jl_amd/csrc EXTRA=-DDH_MI_OCC=$occ > /dev/null 2>&1
  for ff in 0 1; do for cfg in cfg4; do
    run occ${occ}_${cfg}_ff${ff}_wg1 $cfg $ff FW_MI_WG_PER_CU=1
    [ $occ = 2 ] && run occ${occ}_${cfg}_ff${ff}_wg2 $cfg $ff FW_MI_WG_PER_CU=2
  done; done
done
touch flashweave.jl_amd/csrc/fw_devhiton.hip; make -C flashweave.jl_amd/csrc > /dev/null 2>&1
