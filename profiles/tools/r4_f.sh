#!/bin/bash
# r04 run F: fused device round (FW_DH_FUSE 0 / 1 / 2) parity + timing; pipelined level-0 staging
set -u
cd "${GRAFT_REPO_ROOT:-.}"; ROOT=$PWD; export TMPDIR=/tmp
O=$ROOT/gpurun_out/r4_f; rm -rf $O; mkdir -p $O
export FW_KNOBS=1
timeout 900 python -m pytest tests/test_gpu_fz.py -q -x -k "device or rounds or golden or feed" > $O/pytest_fz.txt 2>&1; tail -2 $O/pytest_fz.txt
for f in 0 1 2; do
  FW_DH_FUSE=$f python bench.py --steps 8 --warmup 1 --no-cpu-baseline > $O/cfg3_fuse$f.json 2>/dev/null
done
python - <<PY
import json
for f in (0,1,2):
    d=json.loads(open("$O/cfg3_fuse%d.json"%f).read().strip().splitlines()[-1])
    print("fuse",f,"ms", round(d["ms_per_step"],2), "other", round(d["other_schedule"]["ms_per_step"],2), "one-chain step", round(1e3*d["roofline"]["step_seconds_of_that_pass"],2), "edges", d["edges"], "launches/step", d["kernel_launches_per_step"])
PY
python bench.py --config cfg4 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_cfg4.json 2>/dev/null
python - <<PY
import json
d=json.loads(open("$O/bench_cfg4.json").read().strip().splitlines()[-1])
print("cfg4 ms", round(d["ms_per_step"],2), "other", round(d["other_schedule"]["ms_per_step"],2), "edges", d["edges"], "l0", round(1e3*d["stage_seconds_rank0"]["level0"],2), "cond", round(1e3*d["stage_seconds_rank0"]["conditional"],2))
PY
timeout 1500 python -m pytest tests/test_gpu_mi.py tests/test_gpu_fullsize.py -q -x -k "level0 or cfg3_network_independent or cfg4_full_size_level0 or headline_schedule_device_rounds" > $O/pytest_b.txt 2>&1; tail -2 $O/pytest_b.txt
for n in 8; do
python bench.py --simulate-world 8 --simulate-rank -1 --steps 2 --warmup 1 --no-cpu-baseline --no-other-schedule --no-one-chain > $O/sim8.json 2>/dev/null
FW_DH_FUSE=0 python bench.py --simulate-world 8 --simulate-rank -1 --steps 2 --warmup 1 --no-cpu-baseline --no-other-schedule --no-one-chain > $O/sim8_fuse0.json 2>/dev/null
done
python - <<PY
import json
for f in ("sim8","sim8_fuse0"):
    d=json.loads(open("$O/%s.json"%f).read().strip().splitlines()[-1])
    print(f, "slowest rank ms", round(d["ms_per_step"],2), [round(x,1) for x in d["simulated_world"]["ms_per_step_by_rank"]])
PY
