#!/bin/bash
# r05 run D (re-entry baseline of HEAD): whole pytest -m gpu, then cfg2/cfg3/cfg4/cfg3he lines
O=gpurun_out/r5_d; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $O/pytest.txt
cat $O/pytest.txt
for cfg in cfg2 cfg4 cfg3 cfg3he; do
  timeout 600 python bench.py --config $cfg --steps 5 --warmup 1 --no-cpu-baseline 2>$O/err_$cfg.txt | tail -1 > $O/bench_$cfg.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5_d/bench_*.json")):
    try:
        l=json.loads(open(f).read()); print(f, "ms %.2f other %s edges %d"%(l["ms_per_step"], (l.get("other_schedule") or {}).get("ms_per_step"), l["edges"]), {k:round(v,4) for k,v in l.get("stage_seconds_rank0").items() if k in ("conditional","level0","subsets_kernels_device")}, l["tests_per_step"])
    except Exception as e: print(f, "ERR", e)
PY
