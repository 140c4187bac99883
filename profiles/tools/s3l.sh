timeout 900 python -m pytest tests/test_gpu_fz.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -x -q -k "not cfg5 and not cfg4" 2>&1 | tail -3
python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1), round(d['other_schedule']['ms_per_step'],1), d['edges'], d['kernel_launches_per_step'])"
