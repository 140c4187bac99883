#!/bin/bash
mkdir -p gpurun_out/mipol
run() { # cfg label env...
  cfg=$1; label=$2; shift; shift
  env "$@" FW_TRACE_HOST=1 timeout 300 python bench.py --config $cfg --steps 3 --warmup 1 --feed-forward 0 --no-other-schedule --no-cpu-baseline > gpurun_out/mipol/${cfg}_$label.json 2> gpurun_out/mipol/${cfg}_$label.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/mipol/${cfg}_$label.json").read().strip().splitlines()[-1])
    print("$cfg $label", round(d["ms_per_step"],1), d["edges"], round(d["stage_seconds_rank0"]["subsets_kernels_device"]*1e3,1), d["tests_per_step"]["conditional_evaluated"])
except Exception as e:
    print("$cfg $label no json", e)
PY
  grep -E "per wavefront|watchdog" gpurun_out/mipol/${cfg}_$label.err | tail -1 | cut -c1-260
}
for cfg in cfg2 cfg4; do
run $cfg idle_off FW_MI_IDLE_MIN=0
run $cfg idle_def FW_X=0
run $cfg idle_s1c1 FW_MI_SEQ_IDLE=1 FW_MI_CHUNK_IDLE=1
run $cfg idle_s4c2 FW_MI_SEQ_IDLE=4 FW_MI_CHUNK_IDLE=2
run $cfg idle_s2c4 FW_MI_SEQ_IDLE=2 FW_MI_CHUNK_IDLE=4
run $cfg idle_min64 FW_MI_IDLE_MIN=64
run $cfg idle_min768 FW_MI_IDLE_MIN=768
done
