#!/bin/bash
# discrete persistent kernel, headline schedule (feed-forward rounds): how many targets run as teams (taken dynamically), records, prefixes
cd $GRAFT_REPO_ROOT
export FW_KNOBS=1
run() { name=$1; shift; cfg=$1; shift; ff=$1; shift
  env "$@" timeout 300 python bench.py --config $cfg --feed-forward $ff --steps 3 --warmup 1 --no-cpu-baseline --no-other-schedule --no-one-chain 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$name', round(d['ms_per_step'],2), 'cond', round(1e3*d['stage_seconds_rank0']['conditional'],2), 'eval', d['tests_per_step']['conditional_evaluated'], 'edges', d['edges'])"
}
run cfg4_ff1_off cfg4 1 FW_MI_TEAM_MIN=0
for tm in "64 256" "64 1024" "32 1024" "96 1024" "16 4096"; do set -- $tm
 for ct in 8; do for wt in 128 1024; do for tt in 0 1; do
  run cfg4_ff1_team$1_$2_chunk${ct}_win${wt}_tail$tt cfg4 1 FW_MI_TEAM_MIN=$1 FW_MI_TEAM_MAX=$2 FW_MI_CHUNK_TAIL=$ct FW_MI_WIN0_TAIL=$wt FW_MI_TEAM_TAIL=$tt
 done; done; done
done
run cfg4_ff0_team64_1024 cfg4 0 FW_MI_TEAM_MIN=64 FW_MI_TEAM_MAX=1024 FW_MI_WIN0_TAIL=1024 FW_MI_TEAM_TAIL=1
run cfg2_ff1_team64_1024 cfg2 1 FW_MI_TEAM_MIN=64 FW_MI_TEAM_MAX=1024 FW_MI_WIN0_TAIL=1024 FW_MI_TEAM_TAIL=1
run cfg2_ff1_team32_1024 cfg2 1 FW_MI_TEAM_MIN=32 FW_MI_TEAM_MAX=1024 FW_MI_WIN0_TAIL=1024 FW_MI_TEAM_TAIL=1
run cfg2_ff1_team16_4096 cfg2 1 FW_MI_TEAM_MIN=16 FW_MI_TEAM_MAX=4096 FW_MI_WIN0_TAIL=1024 FW_MI_TEAM_TAIL=1
FW_MI_TEAM_MIN=64 FW_MI_TEAM_MAX=1024 FW_MI_WIN0_TAIL=1024 FW_MI_TEAM_TAIL=1 FW_TRACE_HOST=1 python bench.py --config cfg4 --feed-forward 1 --steps 1 --warmup 0 --no-cpu-baseline --no-other-schedule --no-one-chain 2>&1 >/dev/null | grep -E "finished at|boards " | tail -7 | cut -c1-330
