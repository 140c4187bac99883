cd $GRAFT_REPO_ROOT
export FW_KNOBS=1
timeout 900 python -m pytest tests/test_gpu_mi.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -1
FW_MI_ROW4=0 timeout 900 python -m pytest tests/test_gpu_mi.py -x -q 2>&1 | tail -1
run() { name=$1; shift; cfg=$1; shift; ff=$1; shift; sw=$1; shift
  env "$@" timeout 300 python bench.py --config $cfg --feed-forward $ff $sw --steps 3 --warmup 1 --no-cpu-baseline --no-other-schedule --no-one-chain 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$name', round(d['ms_per_step'],2), 'cond', round(1e3*d['stage_seconds_rank0']['conditional'],2), 'eval', d['tests_per_step']['conditional_evaluated'], 'edges', d['edges'])"
}
run cfg4_ff1 cfg4 1 ""
run cfg4_ff0 cfg4 0 ""
run cfg4_ff1_rank6of8 cfg4 1 "--simulate-world 8 --simulate-rank 6"
run cfg2_ff1 cfg2 1 ""
