#!/bin/bash
# cfg5 at full size: what the slowest of 8 ranks costs (one GPU timing every rank's share in turn; one pass each), headline schedule
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3_cfg5_sim8; mkdir -p $O; cd $R
timeout 2400 python bench.py --config cfg5 --simulate-world 8 --simulate-rank -1 --steps 1 --warmup 0 --no-cpu-baseline --no-other-schedule > $O/cfg5_n8.json 2> $O/cfg5_n8.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r3_cfg5_sim8/cfg5_n8.json") if l.startswith("{")][-1])
sw = d["simulated_world"]
print("cfg5 N=8 slowest rank s", d["ms_per_step"] / 1e3, [round(v / 1e3, 2) for v in sw["ms_per_step_by_rank"]], [round(v, 2) for v in sw["conditional_s_by_rank"]], "edges", d["edges"])
PY
