export FW_KNOBS=1  # the library reads FW_* knobs only when this is set
run() { echo -n "$* : "; env "$@" python bench.py --simulate-world 8 --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1), round(d['other_schedule']['ms_per_step'],1), d['edges'], d['tests_per_step']['conditional_evaluated'])"; }
run A=0
run FW_DH_SPEC=8
run FW_DH_SPEC=16
run FW_DH_SPEC0=4
run FW_DH_SPEC0=8
run FW_DH_SPEC=8 FW_DH_SPEC0=4
run FW_DH_SPEC=16 FW_DH_SPEC0=8 FW_DH_SPEC0_JOBS=2048
run FW_DH_GROWTH=16
run FW_DH_GROWTH=64
