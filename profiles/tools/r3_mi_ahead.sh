#!/bin/bash
# discrete persistent kernel (R4 form): first tests of four candidates per step (mi_first4) -- parity, then cfg2 timings
cd $GRAFT_REPO_ROOT
export FW_KNOBS=1
timeout 900 python -m pytest tests/test_gpu_mi.py tests/test_gpu_fuzz.py tests/test_gpu_dist.py -x -q 2>&1 | tail -2
FW_MI_TEAM_MIN=0 timeout 900 python -m pytest tests/test_gpu_mi.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -1
FW_MI_ROW4=2 FW_MI_TEAM_MIN=0 timeout 900 python -m pytest tests/test_gpu_mi.py -x -q 2>&1 | tail -1
run() { name=$1; shift; cfg=$1; shift; ff=$1; shift
  env "$@" timeout 300 python bench.py --config $cfg --feed-forward $ff --steps 5 --warmup 1 --no-cpu-baseline --no-other-schedule --no-one-chain 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$name', round(d['ms_per_step'],2), 'cond', round(1e3*d['stage_seconds_rank0']['conditional'],2), 'eval', d['tests_per_step']['conditional_evaluated'], 'ref', d['tests_per_step']['conditional_ref_equivalent'], 'edges', d['edges'])"
}
for ah in 0 1; do for tm in 0 64; do
  run cfg2_ahead${ah}_team$tm cfg2 1 FW_MI_AHEAD=$ah FW_MI_TEAM_MIN=$tm
done; done
run cfg2_ahead1_team64_256 cfg2 1 FW_MI_AHEAD=1 FW_MI_TEAM_MIN=128 FW_MI_TEAM_MAX=128
