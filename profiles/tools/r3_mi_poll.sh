cd $GRAFT_REPO_ROOT
export FW_KNOBS=1
timeout 900 python -m pytest tests/test_gpu_mi.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -1
run() { name=$1; shift; cfg=$1; shift; ff=$1; shift
  env "$@" timeout 300 python bench.py --config $cfg --feed-forward $ff --steps 5 --warmup 1 --no-cpu-baseline --no-other-schedule --no-one-chain 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$name', round(d['ms_per_step'],2), 'cond', round(1e3*d['stage_seconds_rank0']['conditional'],2), 'eval', d['tests_per_step']['conditional_evaluated'], 'edges', d['edges'])"
}
for tm in "0 256" "64 256" "128 128" "192 64"; do set -- $tm
  run cfg2_team$1_$2 cfg2 1 FW_MI_TEAM_MIN=$1 FW_MI_TEAM_MAX=$2
done
run cfg4_ff1 cfg4 1 A=1
run cfg4_ff0 cfg4 0 A=1
FW_TRACE_HOST=1 FW_MI_TEAM_MIN=0 python bench.py --config cfg2 --steps 1 --warmup 1 --no-cpu-baseline --no-other-schedule --no-one-chain 2>&1 >/dev/null | grep -E "finished at|boards " | tail -4 | cut -c1-300
