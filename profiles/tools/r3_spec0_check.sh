cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_fz.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -2
for i in 1 2; do python bench.py --steps 10 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1), d['edges'], round(d['other_schedule']['ms_per_step'],1), d['other_schedule']['edges'])"; done
python bench.py --config cfg5 --max-targets 40000 --steps 1 --warmup 0 --no-cpu-baseline --no-other-schedule --no-one-chain 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 first 40000', round(d['ms_per_step'],1), d['edges'])"
