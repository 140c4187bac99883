run() { echo -n "$* : "; env "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1), round(d['other_schedule']['ms_per_step'],1), d['edges'])"; }
run A=0
run FW_SEG_GRID=448
run FW_SEG_GRID=512
run FW_SEG_GRID=768
run FW_SEG_GRID=960
run FW_SEG_GRID=1536
run5() { echo -n "cfg5 p=12000 $* : "; env "$@" python bench.py --config cfg5 --p 12000 --steps 1 --warmup 1 --no-cpu-baseline --no-other-schedule 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1), d['edges'], d['tests_per_step']['conditional_evaluated'])"; }
run5 A=0
run5 FW_SEG_GRID=448
run5 FW_SEG_GRID=960
