#!/bin/bash
# discrete persistent kernel: heavy targets run by a workgroup each (dh_mi_team) -- parity with the team path forced on, then timings
cd $GRAFT_REPO_ROOT
export FW_KNOBS=1
echo "== parity, teams forced (every leading target with >= 2 candidates)"
FW_MI_TEAM_MIN=2 FW_MI_TEAM_MAX=256 timeout 900 python -m pytest tests/test_gpu_mi.py tests/test_gpu_fuzz.py tests/test_gpu_dist.py -x -q 2>&1 | tail -3
FW_MI_TEAM_MIN=2 FW_MI_TEAM_MAX=256 FW_MI_TEAM_STEPS=1 timeout 900 python -m pytest tests/test_gpu_mi.py -x -q 2>&1 | tail -1
echo "== parity, default"
timeout 900 python -m pytest tests/test_gpu_mi.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -1
run() { name=$1; shift; cfg=$1; shift; ff=$1; shift
  env "$@" timeout 300 python bench.py --config $cfg --feed-forward $ff --steps 3 --warmup 1 --no-cpu-baseline --no-other-schedule --no-one-chain 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$name', round(d['ms_per_step'],2), 'cond', round(1e3*d['stage_seconds_rank0']['conditional'],2), 'eval', d['tests_per_step']['conditional_evaluated'], 'edges', d['edges'])"
}
for cfg in cfg4 cfg2; do for ff in 1 0; do
  run ${cfg}_ff${ff}_off $cfg $ff FW_MI_TEAM_MIN=0
  run ${cfg}_ff${ff}_team64 $cfg $ff FW_MI_TEAM_MIN=64
  run ${cfg}_ff${ff}_team32 $cfg $ff FW_MI_TEAM_MIN=32 FW_MI_TEAM_MAX=128
  run ${cfg}_ff${ff}_team64_s4 $cfg $ff FW_MI_TEAM_MIN=64 FW_MI_TEAM_STEPS=4
done; done
FW_TRACE_HOST=1 python bench.py --config cfg4 --steps 1 --warmup 0 --no-cpu-baseline --no-other-schedule --no-one-chain 2>&1 >/dev/null | grep -E "finished at|run by a workgroup|boards " | tail -9
