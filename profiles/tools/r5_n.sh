#!/bin/bash
# r05 run N: the jobs' stop words in device memory (a stop found by one workgroup ends the later segments of its job early):
# fz parity (tests/test_gpu_fz.py, fuzz, cfg3 schedule independence), cfg3 A/B with FW_DH_GSTOP=0, cfg5 sample
O=gpurun_out/r5_n; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_fz.py tests/test_gpu_fuzz.py -q -x 2>&1 | grep -E "passed|failed" > $O/pytest.txt
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -q -x -k "cfg3_network_independent or heavy_tail or cfg3_full_size_headline_schedule_device" 2>&1 | grep -E "passed|failed" >> $O/pytest.txt
cat $O/pytest.txt
for g in 1 0 1 0; do
  FW_KNOBS=1 FW_DH_GSTOP=$g timeout 600 python bench.py --config cfg3 --steps 8 --warmup 2 --no-cpu-baseline --no-one-chain 2>/dev/null | tail -1 > $O/bench_cfg3_g$g.json
  python - <<PY
import json
l=json.loads(open("gpurun_out/r5_n/bench_cfg3_g$g.json").read()); print("gstop=$g cfg3 ms %.2f other %.2f edges %d"%(l["ms_per_step"], l["other_schedule"]["ms_per_step"], l["edges"]), l["tests_per_step"])
PY
done
