O=gpurun_out/r2e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mi.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -5
python profiles/tools/mi_micro.py 1 2 3
FW_TRACE_HOST=1 timeout 300 python bench.py --config cfg4 --steps 1 --warmup 1 --feed-forward 0 --no-other-schedule --no-cpu-baseline 2>&1 | grep "device rounds chain\|conditional stage" | tail -2
timeout 300 python bench.py --config cfg4 --steps 3 --warmup 1 --feed-forward 0 --no-other-schedule --no-cpu-baseline > $O/cfg4.json 2>$O/cfg4.err
FW_DEV_MIN_TARGETS=64 timeout 300 python bench.py --config cfg2 --steps 3 --warmup 1 --feed-forward 0 --no-other-schedule --no-cpu-baseline > $O/cfg2.json 2>$O/cfg2.err
FW_MI_ROUNDS=1 timeout 300 python bench.py --config cfg4 --steps 3 --warmup 1 --feed-forward 0 --no-other-schedule --no-cpu-baseline > $O/cfg4r.json 2>$O/cfg4r.err
timeout 300 python bench.py --config cfg2 --steps 3 --warmup 1 --feed-forward 0 --no-other-schedule --no-cpu-baseline > $O/cfg2h.json 2>$O/cfg2h.err
python - <<'PY'
import json
for f in ['cfg4','cfg2','cfg4r','cfg2h']:
    try:
        j=json.loads([l for l in open('gpurun_out/r2e/%s.json'%f) if l.startswith('{')][-1])
        print(f, round(j['ms_per_step'],2), j['edges'], j['tests_per_step'], {k:round(v,4) for k,v in j['stage_seconds_rank0'].items() if v}, j['roofline']['avg_launch_us'])
    except Exception as e: print(f, 'ERR', e)
PY
