#!/bin/bash
# one rank of eight, every rank in turn (bench.py --simulate-world 8 --simulate-rank -1), plus a per-round log of rank 0
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3_sim8
mkdir -p $O
cd $R
timeout 900 python bench.py --simulate-world 8 --simulate-rank -1 --steps 3 --warmup 1 --no-cpu-baseline > $O/sim8.json 2> $O/sim8.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3_sim8/sim8.json").read().strip().splitlines()[-1])
print("ff=1", d["ms_per_step"], d["simulated_world"]["ms_per_step_by_rank"], d["simulated_world"]["cond_ref_by_rank"])
o = d["other_schedule"]
print("ff=0", o["ms_per_step"], o["simulated_world"]["ms_per_step_by_rank"], o["simulated_world"]["cond_ref_by_rank"])
PY
FW_TRACE_HOST=1 FW_DH_LOG=$O/dhlog_rank0.txt FW_DH_CHAINS=1 timeout 600 python bench.py --simulate-world 8 --simulate-rank 0 --steps 1 --warmup 0 --no-cpu-baseline --no-other-schedule > $O/trace0.json 2> $O/trace0.err
tail -60 $O/trace0.err
