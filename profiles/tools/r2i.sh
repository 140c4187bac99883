FW_DEV_MIN_TARGETS=1 FW_TRACE_HOST=1 timeout 60 python profiles/tools/dbg_mi.py 2>&1 | grep -v amdgpu | tail -3
timeout 300 python -m pytest tests/test_gpu_mi.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -3
run() { echo "== $*"; env "$@" FW_TRACE_HOST=1 timeout 60 python bench.py --config ${CFG:-cfg4} --steps 2 --warmup 1 --feed-forward 0 --no-other-schedule --no-cpu-baseline 2>&1 | grep "device rounds chain\|boards" | tail -2; }
run A=1
run FW_MI_SEQ=2 FW_MI_WIN0=32 FW_MI_CHUNK_MIN=2
run FW_MI_SEQ=8 FW_MI_WIN0=128 FW_MI_CHUNK_MIN=4
run FW_MI_SEQ=4 FW_MI_WIN0=64 FW_MI_CHUNK_MIN=2 FW_MI_CHUNK_DIV=1024
run FW_MI_SEQ=16 FW_MI_WIN0=128 FW_MI_CHUNK_MIN=8
CFG=cfg2
run A=1
run FW_MI_SEQ=2 FW_MI_WIN0=32 FW_MI_CHUNK_MIN=2
