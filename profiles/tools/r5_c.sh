#!/bin/bash
# r05 run C: (1) first tests of 16 candidates per team round + Q(1/2, x) through erfc: discrete parity, cfg2 / cfg4; (2) the p = 0 regime of
# the maximum-p bookkeeping: fz parity, cfg5 at full size, cfg3
O=gpurun_out/r5_c; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_mi.py tests/test_gpu_fuzz.py tests/test_gpu_fz.py -q -x 2>&1 | tail -5 > $O/pytest.txt
timeout 1500 python -m tests.fuzz_gpu --subsets --first 920000 --cases 600 2>&1 | tail -2 >> $O/pytest.txt
timeout 1500 python -m tests.fuzz_gpu --first 930000 --cases 600 2>&1 | tail -2 >> $O/pytest.txt
timeout 2400 python -m pytest tests/test_gpu_fullsize.py -q -x -k "cfg5 or cfg4 or heavy_tail" 2>&1 | tail -4 >> $O/pytest.txt
cat $O/pytest.txt
timeout 600 python bench.py --config cfg2 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_cfg2.json
timeout 600 python bench.py --config cfg4 --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_cfg4.json
timeout 600 python bench.py --config cfg3 --steps 8 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_cfg3.json
timeout 900 python bench.py --config cfg5 --steps 1 --warmup 0 --no-cpu-baseline --no-other-schedule --no-one-chain 2>$O/err_cfg5.txt | tail -1 > $O/bench_cfg5.json
FW_KNOBS=1 FW_TRACE_HOST=1 timeout 600 python bench.py --config cfg4 --steps 1 --warmup 1 --no-cpu-baseline --no-other-schedule --no-one-chain 2> $O/cfg4_trace.txt >/dev/null
FW_KNOBS=1 FW_TRACE_HOST=1 timeout 600 python bench.py --config cfg2 --steps 1 --warmup 1 --no-cpu-baseline --no-other-schedule --no-one-chain 2> $O/cfg2_trace.txt >/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5_c/bench_*.json")):
    try:
        l=json.loads(open(f).read()); print(f, "ms %.2f other %s edges %d"%(l["ms_per_step"], (l.get("other_schedule") or {}).get("ms_per_step"), l["edges"]), {k:round(v,4) for k,v in l.get("stage_seconds_rank0").items() if k in ("conditional","level0","subsets_kernels_device")}, l["tests_per_step"])
    except Exception as e: print(f, "ERR", e)
PY
grep "device schedule\|weights\|neighbour lists\|conditional stage" $O/cfg4_trace.txt | tail -4
grep "finished at\|boards\|device schedule" $O/cfg2_trace.txt | tail -5 | cut -c1-250
