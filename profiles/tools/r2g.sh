timeout 300 python -m pytest tests/test_gpu_mi.py -x -q 2>&1 | tail -2
run() { echo "== $*"; env "$@" FW_TRACE_HOST=1 timeout 60 python bench.py --config ${CFG:-cfg4} --steps 2 --warmup 1 --feed-forward 0 --no-other-schedule --no-cpu-baseline 2>&1 | grep "device rounds chain" | tail -1; }
run A=1
run FW_MI_HELP_JOBS=0
run FW_MI_SEQ=8 FW_MI_WIN0=64
run FW_MI_SEQ=8 FW_MI_WIN0=64 FW_MI_CHUNK_MIN=4
run FW_MI_SEQ=16 FW_MI_WIN0=512 FW_MI_CHUNK_MIN=4 FW_MI_CHUNK_DIV=1024
run FW_MI_WG_PER_CU=1
CFG=cfg2; export FW_DEV_MIN_TARGETS=64
run A=1
run FW_MI_SEQ=8 FW_MI_WIN0=64 FW_MI_CHUNK_MIN=4
