#!/bin/bash
export FW_KNOBS=1  # the library reads FW_* knobs only when this is set
# one rank of eight (rank 3: the slowest of r3_sim8), knob sweep + where the longest-busy target spends its rounds
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3_sim8b
mkdir -p $O
cd $R
run() { # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --simulate-world 8 --simulate-rank 3 --steps 3 --warmup 1 --no-cpu-baseline --no-other-schedule > $O/$name.json 2> $O/$name.err
  python - "$name" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r3_sim8b/%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d["ms_per_step"], 2), "cond", round(1e3 * d["stage_seconds_rank0"]["conditional"], 2), "eval", d["tests_per_step"]["conditional_evaluated"])
PY
}
run base A=1
run small16M FW_SMALL_LAUNCH=16000000
run small64M FW_SMALL_LAUNCH=64000000
run growth16 FW_DH_GROWTH=16 FW_DH_GROWTH_BUSY=16
run w0big64k FW_W0_BIG=65536
run w0big1M FW_W0_BIG=1048576
run spec0_8 FW_DH_SPEC0=8
run spec8 FW_DH_SPEC=8
run chains1 FW_DH_CHAINS=1
run all FW_SMALL_LAUNCH=64000000 FW_W0_BIG=262144 FW_DH_SPEC0=8 FW_DH_SPEC=8
FW_TRACE_HOST=1 timeout 300 python bench.py --simulate-world 8 --simulate-rank 3 --steps 1 --warmup 0 --no-cpu-baseline --no-other-schedule > $O/trace.json 2> $O/trace.err
grep "longest-busy\|device rounds chain" $O/trace.err | tail -8
