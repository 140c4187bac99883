#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3_sim8c
mkdir -p $O
cd $R
run() { # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --simulate-world 8 --simulate-rank 3 --steps 4 --warmup 1 --no-cpu-baseline --no-other-schedule > $O/$name.json 2> $O/$name.err
  python - "$name" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r3_sim8c/%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d["ms_per_step"], 2), "cond", round(1e3 * d["stage_seconds_rank0"]["conditional"], 2), "eval", d["tests_per_step"]["conditional_evaluated"])
PY
}
run base A=1
run base2 A=1
run cmin48_2 FW_DH_CHAIN_MIN=48 FW_DH_CHAINS=2
run cmin32_3 FW_DH_CHAIN_MIN=32 FW_DH_CHAINS=3
run cmin24_4 FW_DH_CHAIN_MIN=24 FW_DH_CHAINS=4
run cmin48_2_g16 FW_DH_CHAIN_MIN=48 FW_DH_CHAINS=2 FW_DH_GROWTH=16
run cmin32_3_g16 FW_DH_CHAIN_MIN=32 FW_DH_CHAINS=3 FW_DH_GROWTH=16
run cmin32_3_spec1_4 FW_DH_CHAIN_MIN=32 FW_DH_CHAINS=3 FW_DH_SPEC1=4
run spec1_0 FW_DH_SPEC1=0
