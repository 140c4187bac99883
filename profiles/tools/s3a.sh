O=gpurun_out/s3a; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_fz.py tests/test_gpu_fznz.py -x -q > $O/pytest_fz.txt 2>&1; echo rc=$? >> $O/pytest_fz.txt
timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_cfg3.json 2> $O/bench_cfg3.err; echo rc=$?
tail -3 $O/pytest_fz.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/s3a/bench_cfg3.json'))
print(d['ms_per_step'], d['value'], d['edges'], d['other_schedule']['ms_per_step'], d['other_schedule']['edges'], d['roofline']['frac'], d['roofline']['avg_launch_us'])
PY
