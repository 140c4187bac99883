run() { echo "== $*"; env "$@" FW_TRACE_HOST=1 timeout 300 python bench.py --config cfg4 --steps 1 --warmup 1 --feed-forward 0 --no-other-schedule --no-cpu-baseline 2>&1 | grep "device rounds chain" | tail -1; }
run A=1
run FW_MI_HELP_JOBS=0
run FW_MI_SEQ=64 FW_MI_WIN0=512
run FW_MI_SEQ=64 FW_MI_WIN0=512 FW_MI_HELP_JOBS=0
run FW_MI_SEQ=16 FW_MI_WIN0=1024 FW_MI_CHUNK_MIN=16
run FW_MI_SEQ=1000000000
