O=gpurun_out/r2d; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_mi.py -x -q 2>&1 | tail -3
FW_TRACE_HOST=1 FW_DH_CHAINS_DISC=1 timeout 300 python bench.py --config cfg4 --steps 1 --warmup 1 --feed-forward 0 --no-other-schedule --no-cpu-baseline 2>&1 | grep "device rounds chain\|conditional stage" | tail -2
timeout 300 python bench.py --config cfg4 --steps 3 --warmup 1 --feed-forward 0 --no-other-schedule --no-cpu-baseline > $O/cfg4.json 2>$O/cfg4.err
FW_DEV_MIN_TARGETS=64 timeout 300 python bench.py --config cfg2 --steps 3 --warmup 1 --feed-forward 0 --no-other-schedule --no-cpu-baseline > $O/cfg2.json 2>$O/cfg2.err
python - <<'PY'
import json
for f in ['cfg4','cfg2']:
    try:
        j=json.loads([l for l in open('gpurun_out/r2d/%s.json'%f) if l.startswith('{')][-1])
        print(f, round(j['ms_per_step'],2), j['edges'], j['tests_per_step'], {k:round(v,4) for k,v in j['stage_seconds_rank0'].items() if v}, j['roofline']['avg_launch_us'])
    except Exception as e: print(f, 'ERR', e)
PY
tail -3 $O/cfg4.err
