#!/bin/bash
# r04: row form of the discrete counting phase (mi_bin_rows) -- parity tests, micro-benchmark per k, cfg2 / cfg4 passes; A/B by FW_MI_ROWK
O=gpurun_out/r4_k; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mi.py tests/test_gpu_fuzz.py -q -x 2>&1 | tail -8 > $O/pytest.txt
for rk in 99 3 2 1; do
  FW_KNOBS=1 FW_MI_ROWK=$rk timeout 300 python profiles/tools/mi_micro.py 1 2 3 2>&1 | grep "k=" | sed "s/^/rowk=$rk mi_nz /" >> $O/micro.txt
  KIND=mi N=500 FW_KNOBS=1 FW_MI_ROWK=$rk timeout 300 python profiles/tools/mi_micro.py 1 2 3 2>&1 | grep "k=" | sed "s/^/rowk=$rk mi n500 /" >> $O/micro.txt
done
for rk in 99 3 2; do
  for cfg in cfg2 cfg4; do
    FW_KNOBS=1 FW_MI_ROWK=$rk timeout 600 python bench.py --config $cfg --steps 5 --warmup 1 --no-cpu-baseline 2>$O/err_${cfg}_$rk.txt | tail -1 > $O/bench_${cfg}_rowk$rk.json
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4_k/bench_*.json")):
    try:
        l=json.loads(open(f).read()); print(f, "ms %.2f other %.2f edges %d"%(l["ms_per_step"], l["other_schedule"]["ms_per_step"], l["edges"]))
    except Exception as e: print(f, "ERR", e)
PY
cat $O/pytest.txt $O/micro.txt
