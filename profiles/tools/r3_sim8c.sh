#!/bin/bash
export FW_KNOBS=1  # the library reads FW_* knobs only when this is set
# one rank of eight (cfg3, headline schedule): the heaviest targets in a chain of their own
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3_sim8c
mkdir -p $O
cd $R
run() { name=$1; rk=$2; shift; shift
  env "$@" timeout 300 python bench.py --simulate-world 8 --simulate-rank $rk --steps 4 --warmup 1 --no-cpu-baseline --no-other-schedule > $O/$name.json 2> $O/$name.err
  python - "$name" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r3_sim8c/%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d["ms_per_step"], 2), "cond", round(1e3 * d["stage_seconds_rank0"]["conditional"], 2), "eval", d["tests_per_step"]["conditional_evaluated"], "edges", d["edges"])
PY
}
for rk in 2 7; do
run r${rk}_base $rk A=1
run r${rk}_h10 $rk FW_DH_HEAVY_PCT=10
run r${rk}_h20 $rk FW_DH_HEAVY_PCT=20
run r${rk}_h35 $rk FW_DH_HEAVY_PCT=35
run r${rk}_h20_c3 $rk FW_DH_HEAVY_PCT=20 FW_DH_CHAINS=3 FW_DH_CHAIN_MIN=32
run r${rk}_h10_c3 $rk FW_DH_HEAVY_PCT=10 FW_DH_CHAINS=3 FW_DH_CHAIN_MIN=32
done
