#!/bin/bash
export FW_KNOBS=1  # the library reads FW_* knobs only when this is set
# discrete persistent kernel: owner prefix in the tail (FW_MI_SEQ_TAIL), one rank of eight (rank 6) and the single-rank pass, cfg4 / cfg2
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3_mi_knobs
mkdir -p $O
cd $R
run() { name=$1; cfg=$2; sw=$3; shift; shift; shift
  env "$@" timeout 300 python bench.py --config $cfg $sw --feed-forward 0 --steps 3 --warmup 1 --no-cpu-baseline --no-other-schedule > $O/$name.json 2> $O/$name.err
  python - "$name" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r3_mi_knobs/%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d["ms_per_step"], 2), "cond", round(1e3 * d["stage_seconds_rank0"]["conditional"], 2), "l0", round(1e3 * d["stage_seconds_rank0"]["level0"], 2), "eval", d["tests_per_step"]["conditional_evaluated"], "edges", d["edges"])
PY
}
S8="--simulate-world 8 --simulate-rank 6"
for T in 16 4; do run t${T}_8 cfg4 "$S8" FW_MI_SEQ_TAIL=$T; done
for T in 16 4; do run t${T}_1 cfg4 "" FW_MI_SEQ_TAIL=$T; done
for T in 16 4 2; do run c2_t${T} cfg2 "" FW_MI_SEQ_TAIL=$T; done
