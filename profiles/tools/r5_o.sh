#!/bin/bash
# r05 run O: cfg5 at full size with / without the jobs' stop words
O=gpurun_out/r5_o; mkdir -p $O
for g in 1 0; do
  FW_KNOBS=1 FW_DH_GSTOP=$g timeout 900 python bench.py --config cfg5 --steps 1 --warmup 0 --no-cpu-baseline --no-other-schedule --no-one-chain 2>$O/err_g$g.txt | tail -1 > $O/bench_cfg5_g$g.json
  python - <<PY
import json
l=json.loads(open("gpurun_out/r5_o/bench_cfg5_g$g.json").read()); print("gstop=$g cfg5 s %.2f edges %d"%(l["ms_per_step"]/1e3, l["edges"]), l["tests_per_step"], {k:round(v,3) for k,v in l["stage_seconds_rank0"].items() if k in ("conditional","level0","subsets_kernels_device")})
PY
done
