O=gpurun_out/s3b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fz.py tests/test_gpu_fznz.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -x -q -k "not cfg5 and not cfg4" > $O/pytest_fz.txt 2>&1; echo rc=$? >> $O/pytest_fz.txt
timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_cfg3.json 2> $O/bench_cfg3.err; echo rc=$?
tail -3 $O/pytest_fz.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/s3b/bench_cfg3.json'))
print(d['ms_per_step'], d['value'], d['edges'], d['other_schedule']['ms_per_step'], d['other_schedule']['edges'], d['roofline']['frac'], d['roofline']['avg_launch_us'])
PY
