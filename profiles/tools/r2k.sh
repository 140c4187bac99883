timeout 300 python -m pytest tests/test_gpu_mi.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -2
run() { echo "== $*"; env "$@" FW_TRACE_HOST=1 timeout 60 python bench.py --config ${CFG:-cfg4} --steps 3 --warmup 1 --feed-forward 0 --no-other-schedule --no-cpu-baseline 2>&1 | grep "device rounds chain" | tail -2; }
run A=1
run FW_MI_SEQ=12 FW_MI_WIN0=96 FW_MI_CHUNK_MIN=6
run FW_MI_SEQ=24 FW_MI_WIN0=256 FW_MI_CHUNK_MIN=8
CFG=cfg2
run A=1
