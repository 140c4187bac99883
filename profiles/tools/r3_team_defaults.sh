cd $GRAFT_REPO_ROOT
export FW_KNOBS=1
run() { name=$1; shift; cfg=$1; shift; ff=$1; shift; sw=$1; shift
  env "$@" timeout 300 python bench.py --config $cfg --feed-forward $ff $sw --steps 4 --warmup 1 --no-cpu-baseline --no-other-schedule --no-one-chain 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$name', round(d['ms_per_step'],2), 'cond', round(1e3*d['stage_seconds_rank0']['conditional'],2), 'eval', d['tests_per_step']['conditional_evaluated'], 'edges', d['edges'])"
}
for tm in "64 256" "128 128" "128 256" "96 192"; do set -- $tm
  run cfg4_ff1_team$1_$2 cfg4 1 "" FW_MI_TEAM_MIN=$1 FW_MI_TEAM_MAX=$2
  run cfg4_ff0_team$1_$2 cfg4 0 "" FW_MI_TEAM_MIN=$1 FW_MI_TEAM_MAX=$2
  run cfg4_ff1_r6of8_team$1_$2 cfg4 1 "--simulate-world 8 --simulate-rank 6" FW_MI_TEAM_MIN=$1 FW_MI_TEAM_MAX=$2
  run cfg2_team$1_$2 cfg2 1 "" FW_MI_TEAM_MIN=$1 FW_MI_TEAM_MAX=$2
done
