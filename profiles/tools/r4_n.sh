#!/bin/bash
# r04: validation of the matrix-core level-0 kernel on the final build: discrete tests, sharded forms, fuzz, cfg4 at full size against
# the oracle; cfg4 / cfg2 with and without it
O=gpurun_out/r4_n; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_mi.py tests/test_gpu_dist.py tests/test_gpu_fuzz.py -q 2>&1 | grep -i "passed\|failed\|error" | tail -8 > $O/pytest.txt
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -k "cfg4" 2>&1 | grep -i "passed\|failed\|error" | tail -8 >> $O/pytest.txt
for m in 0 1; do
  FW_KNOBS=1 FW_L0_MFMA=$m timeout 600 python bench.py --config cfg4 --steps 5 --warmup 1 --no-cpu-baseline 2>$O/err_$m.txt | tail -1 > $O/bench_cfg4_mfma$m.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4_n/bench_*.json")):
    try:
        l=json.loads(open(f).read()); print(f, "ms %.2f other %.2f edges %d"%(l["ms_per_step"], l["other_schedule"]["ms_per_step"], l["edges"]), {k:round(v,4) for k,v in l.get("stage_seconds_rank0").items() if k in("level0","conditional")})
    except Exception as e: print(f, "ERR", e)
PY
cat $O/pytest.txt
