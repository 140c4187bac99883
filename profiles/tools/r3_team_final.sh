#!/bin/bash
# discrete persistent kernel with team targets (defaults): parity (teams forced / default), one rank of eight, headline lines
cd $GRAFT_REPO_ROOT
export FW_KNOBS=1
FW_MI_TEAM_MIN=2 FW_MI_TEAM_MAX=100000 timeout 900 python -m pytest tests/test_gpu_mi.py tests/test_gpu_fuzz.py tests/test_gpu_dist.py -x -q 2>&1 | tail -2
FW_MI_TEAM_MIN=2 FW_MI_TEAM_MAX=100000 FW_MI_TEAM_TAIL=1 FW_MI_CHUNK_TAIL=1 timeout 900 python -m pytest tests/test_gpu_mi.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -1
timeout 1200 python -m pytest tests/test_gpu_mi.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_dist.py -x -q 2>&1 | tail -1
run() { name=$1; shift; cfg=$1; shift; ff=$1; shift; sw=$1; shift
  env "$@" timeout 300 python bench.py --config $cfg --feed-forward $ff $sw --steps 3 --warmup 1 --no-cpu-baseline --no-other-schedule --no-one-chain 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$name', round(d['ms_per_step'],2), 'cond', round(1e3*d['stage_seconds_rank0']['conditional'],2), 'eval', d['tests_per_step']['conditional_evaluated'], 'edges', d['edges'])"
}
for tm in 0 64; do
run cfg4_ff1_team$tm cfg4 1 "" FW_MI_TEAM_MIN=$tm
run cfg4_ff0_team$tm cfg4 0 "" FW_MI_TEAM_MIN=$tm
run cfg4_ff1_rank6of8_team$tm cfg4 1 "--simulate-world 8 --simulate-rank 6" FW_MI_TEAM_MIN=$tm
run cfg4_ff0_rank6of8_team$tm cfg4 0 "--simulate-world 8 --simulate-rank 6" FW_MI_TEAM_MIN=$tm
run cfg2_ff1_team$tm cfg2 1 "" FW_MI_TEAM_MIN=$tm
done
