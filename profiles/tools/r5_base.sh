#!/bin/bash
# r05: baseline of the round on this box (r04 code + advisor fixes): cfg2 / cfg4 / cfg3 lines and the host-side trace of cfg4 / cfg2
O=gpurun_out/r5_base; mkdir -p $O
for cfg in cfg2 cfg4 cfg3; do
  timeout 900 python bench.py --config $cfg --steps 5 --warmup 1 --no-cpu-baseline 2>$O/err_$cfg.txt | tail -1 > $O/bench_$cfg.json
done
FW_KNOBS=1 FW_TRACE_HOST=1 timeout 600 python bench.py --config cfg4 --steps 1 --warmup 1 --no-cpu-baseline --no-other-schedule --no-one-chain 2> $O/cfg4_trace.txt >/dev/null
FW_KNOBS=1 FW_TRACE_HOST=1 timeout 600 python bench.py --config cfg2 --steps 1 --warmup 1 --no-cpu-baseline --no-other-schedule --no-one-chain 2> $O/cfg2_trace.txt >/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5_base/bench_*.json")):
    try:
        l=json.loads(open(f).read()); print(f, "ms %.2f other %.2f edges %d"%(l["ms_per_step"], l["other_schedule"]["ms_per_step"], l["edges"]), {k:round(v,4) for k,v in l.get("stage_seconds_rank0").items()})
    except Exception as e: print(f, "ERR", e)
PY
