#!/bin/bash
# r05 run H: level-0 matrix-core kernel, new staging / pre-permuted operands / shifted barrier: parity (level-0 tests, cfg4 rows vs oracle),
# phase cycles, stage seconds with and without the epilogue, cfg4 line
O=gpurun_out/r5_level0; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mi.py tests/test_gpu_dist.py -q -x -k "level0 or matrix_core or levels or network or sharded or rccl" 2>&1 | grep -E "passed|failed" > $O/pytest.txt
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -x -k "cfg4_full_size_level0" 2>&1 | grep -E "passed|failed" >> $O/pytest.txt
cat $O/pytest.txt
for d in 0 1; do
  echo "== FW_L0_DBG=$d" >> $O/l0_ablate.txt
  FW_KNOBS=1 FW_L0_VERBOSE=1 L0_ABLATE_SET="$d" timeout 600 python profiles/tools/l0_ablate.py 2>&1 | grep -E "shader cycles|^[0-9] " | tail -2 >> $O/l0_ablate.txt
done
cat $O/l0_ablate.txt
timeout 600 python bench.py --config cfg4 --steps 5 --warmup 1 --no-cpu-baseline 2>$O/err_cfg4.txt | tail -1 > $O/bench_cfg4.json
python - <<'PY'
import json
l=json.loads(open("gpurun_out/r5_level0/bench_cfg4.json").read()); print("cfg4 ms %.2f other %s edges %d"%(l["ms_per_step"], (l.get("other_schedule") or {}).get("ms_per_step"), l["edges"]), {k:round(v,4) for k,v in l.get("stage_seconds_rank0").items() if k in ("conditional","level0")})
PY
