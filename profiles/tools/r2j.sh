run() { echo "== $*"; env "$@" FW_TRACE_HOST=1 timeout 60 python bench.py --config ${CFG:-cfg4} --steps 2 --warmup 1 --feed-forward 0 --no-other-schedule --no-cpu-baseline 2>&1 | grep "device rounds chain\|boards" | tail -2; }
run FW_MI_SEQ=16 FW_MI_WIN0=128 FW_MI_CHUNK_MIN=8
run FW_MI_SEQ=8 FW_MI_WIN0=128 FW_MI_CHUNK_MIN=4
run A=1
