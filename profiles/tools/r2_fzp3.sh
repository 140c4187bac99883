#!/bin/bash
mkdir -p gpurun_out/fzp
run() { # label, args..., env via PERSIST
  label=$1; shift
  FW_FZ_PERSIST=$PERSIST FW_TRACE_HOST=1 timeout 300 python bench.py --config cfg3 --steps 3 --warmup 1 --no-other-schedule --no-cpu-baseline "$@" > gpurun_out/fzp/$label.json 2> gpurun_out/fzp/$label.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/fzp/$label.json").read().strip().splitlines()[-1])
    print("$label persist=$PERSIST", round(d["ms_per_step"],1), d["edges"], round(d["stage_seconds_rank0"]["subsets_kernels_device"]*1e3,1))
except Exception as e:
    print("$label no json", e)
PY
}
for PERSIST in 0 1; do
  run w8_ff1_$PERSIST --simulate-world 8 --feed-forward 1
  run w8_ff0_$PERSIST --simulate-world 8 --feed-forward 0
  run w2_ff1_$PERSIST --simulate-world 2 --feed-forward 1
done
