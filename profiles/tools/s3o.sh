run() { echo -n "$* : "; env "$@" python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1), round(d['other_schedule']['ms_per_step'],1), d['edges'], d['other_schedule']['edges'], d['tests_per_step']['conditional_evaluated'], d['other_schedule']['tests_per_step']['conditional_evaluated'])"; }
run A=0
run A=0
timeout 900 python -m pytest tests/test_gpu_fz.py tests/test_gpu_fznz.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -2
