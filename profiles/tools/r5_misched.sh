#!/bin/bash
# r05: the discrete kinds' whole schedule on the device -- parity (tests/test_gpu_mi.py, fuzz, cfg4 at full size against the oracle), then
# cfg4 / cfg2 with FW_MI_SCHED = 1 (default) / 0 (per-round loop), and the host-side trace
O=gpurun_out/r5_misched; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_mi.py tests/test_gpu_fuzz.py -q -x 2>&1 | tail -5 > $O/pytest.txt
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -q -x -k "cfg4" 2>&1 | tail -5 >> $O/pytest.txt
cat $O/pytest.txt
for m in 1 0; do
  FW_KNOBS=1 FW_MI_SCHED=$m timeout 600 python bench.py --config cfg4 --steps 5 --warmup 1 --no-cpu-baseline 2>$O/err_$m.txt | tail -1 > $O/bench_cfg4_sched$m.json
  FW_KNOBS=1 FW_MI_SCHED=$m timeout 600 python bench.py --config cfg2 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_cfg2_sched$m.json
done
FW_KNOBS=1 FW_TRACE_HOST=1 timeout 600 python bench.py --config cfg4 --steps 1 --warmup 1 --no-cpu-baseline --no-other-schedule --no-one-chain 2> $O/cfg4_trace.txt >/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5_misched/bench_*.json")):
    try:
        l=json.loads(open(f).read()); print(f, "ms %.2f other %.2f edges %d"%(l["ms_per_step"], l["other_schedule"]["ms_per_step"], l["edges"]), {k:round(v,4) for k,v in l.get("stage_seconds_rank0").items() if k in ("conditional","level0")}, l["tests_per_step"])
    except Exception as e: print(f, "ERR", e)
PY
grep -v "finished at\|test routine\|state machine" $O/cfg4_trace.txt | tail -22 | cut -c1-260
