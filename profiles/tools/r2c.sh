O=gpurun_out/r2c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mi.py -x -q 2>&1 | tail -3
FW_TRACE_HOST=1 FW_DH_CHAINS_DISC=1 python bench.py --config cfg4 --steps 1 --warmup 1 --feed-forward 0 --no-other-schedule --no-cpu-baseline 2>&1 | grep "device rounds chain\|conditional stage" | tail -2
FW_MI_ROUNDS=1 python bench.py --config cfg4 --steps 3 --warmup 1 --feed-forward 0 --no-other-schedule --no-cpu-baseline > $O/cfg4_rounds.json 2>/dev/null
FW_MI_ROUNDS=1 python bench.py --config cfg2 --steps 3 --warmup 1 --feed-forward 0 --no-other-schedule --no-cpu-baseline > $O/cfg2_host.json 2>/dev/null
FW_TRACE_HOST=1 FW_DEV_MIN_TARGETS=64 python bench.py --config cfg2 --steps 1 --warmup 1 --feed-forward 0 --no-other-schedule --no-cpu-baseline 2>&1 | grep "device rounds chain\|conditional stage" | tail -3
python - <<'PY'
import json
for f in ['cfg4_rounds','cfg2_host']:
    j=json.loads([l for l in open('gpurun_out/r2c/%s.json'%f) if l.startswith('{')][-1])
    print(f, round(j['ms_per_step'],2), {k:round(v,4) for k,v in j['stage_seconds_rank0'].items() if v}, j['roofline']['avg_launch_us'])
PY
