#!/bin/bash
mkdir -p gpurun_out/fzp
run() { # label, env...
  label=$1; shift
  env "$@" FW_TRACE_HOST=1 timeout 300 python bench.py --config cfg3 --steps 2 --warmup 1 --feed-forward 0 --no-other-schedule --no-cpu-baseline > gpurun_out/fzp/$label.json 2> gpurun_out/fzp/$label.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/fzp/$label.json").read().strip().splitlines()[-1])
    print("$label", round(d["ms_per_step"],1), d["edges"], round(d["stage_seconds_rank0"]["subsets_kernels_device"]*1e3,1))
except Exception as e:
    print("$label no json", e)
PY
  grep -E "per workgroup|watchdog" gpurun_out/fzp/$label.err | tail -1
}
run base FW_X=0
run wg1 FW_FZ_WG_PER_CU=1
run wg2 FW_FZ_WG_PER_CU=2
run nohelp FW_MI_HELP_JOBS=0
