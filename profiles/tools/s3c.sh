O=gpurun_out/s3c; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_fznz.py -x -q > $O/pytest_fznz.txt 2>&1; echo rc=$? >> $O/pytest_fznz.txt
tail -3 $O/pytest_fznz.txt
timeout 300 python bench.py --config cfg3he --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_cfg3he.json 2> $O/bench_cfg3he.err; echo rc=$?
python - <<'PY'
import json
d=json.load(open('gpurun_out/s3c/bench_cfg3he.json'))
print(d['ms_per_step'], d['value'], d['edges'], d.get('other_schedule',{}).get('ms_per_step'), d['stage_seconds_rank0'])
PY
