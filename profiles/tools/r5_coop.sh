#!/bin/bash
# r05: the cooperative round (dh_coop_kernel) -- parity tests of the fz device rounds, then cfg3 with FW_DH_FUSE = 1 (coop) / 0 (three launches)
O=gpurun_out/r5_coop; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_fz.py tests/test_gpu_fuzz.py -q -x 2>&1 | tail -5 > $O/pytest.txt
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -x -k "cfg3_network_independent or device_rounds_equal_host" 2>&1 | tail -5 >> $O/pytest.txt
cat $O/pytest.txt
for f in 1 0; do
  FW_KNOBS=1 FW_DH_FUSE=$f timeout 600 python bench.py --config cfg3 --steps 8 --warmup 1 --no-cpu-baseline 2>$O/err_$f.txt | tail -1 > $O/bench_cfg3_fuse$f.json
done
FW_KNOBS=1 FW_TRACE_HOST=1 timeout 600 python bench.py --config cfg3 --steps 1 --warmup 1 --no-cpu-baseline --no-other-schedule --no-one-chain 2> $O/cfg3_trace.txt >/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5_coop/bench_*.json")):
    try:
        l=json.loads(open(f).read()); print(f, "ms %.2f other %.2f edges %d"%(l["ms_per_step"], l["other_schedule"]["ms_per_step"], l["edges"]), l["roofline"].get("one_chain_step_ms"), {k:round(v,4) for k,v in l.get("stage_seconds_rank0").items() if k in ("conditional","level0")})
    except Exception as e: print(f, "ERR", e)
PY
