#!/bin/bash
mkdir -p gpurun_out/spec
run() { # label, world, env...
  label=$1; w=$2; shift; shift
  env "$@" timeout 300 python bench.py --config cfg3 --steps 4 --warmup 1 --no-other-schedule --no-cpu-baseline --simulate-world $w --feed-forward 1 > gpurun_out/spec/$label.json 2> gpurun_out/spec/$label.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/spec/$label.json").read().strip().splitlines()[-1])
    print("$label", round(d["ms_per_step"],1), d["edges"], d["tests_per_step"]["conditional_evaluated"], d.get("kernel_launches_per_step"))
except Exception as e:
    print("$label no json", e)
PY
}
for w in 8 1; do
run w${w}_base $w FW_X=0
run w${w}_spec8 $w FW_DH_SPEC=8
run w${w}_spec0_4 $w FW_DH_SPEC0=4
run w${w}_spec0_8 $w FW_DH_SPEC0=8 FW_DH_SPEC=8
run w${w}_chains3 $w FW_DH_CHAINS=3
run w${w}_chains1 $w FW_DH_CHAINS=1
done
