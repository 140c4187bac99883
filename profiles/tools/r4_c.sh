#!/bin/bash
# r04 run C: transposed gathers (CORT) + level-0 2x2 bound: parity quick checks and timing
set -u
cd "${GRAFT_REPO_ROOT:-.}"; ROOT=$PWD; export TMPDIR=/tmp
O=$ROOT/gpurun_out/r4_c; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fz.py tests/test_gpu_mi.py -q -x > $O/pytest_a.txt 2>&1; tail -3 $O/pytest_a.txt
python bench.py --steps 10 --warmup 1 --no-cpu-baseline > $O/bench_cfg3.json 2> $O/bench_cfg3.err
python bench.py --config cfg4 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench_cfg4.err
python - <<PY
import json
for c in ("cfg3","cfg4"):
    d=json.loads(open("$O/bench_%s.json"%c).read().strip().splitlines()[-1])
    print(c,"ms", d["ms_per_step"], "other", d["other_schedule"]["ms_per_step"], "frac", d["roofline"]["frac"], "edges", d["edges"], "eval", d["tests_per_step"]["conditional_evaluated"], "stages", d["stage_seconds_rank0"]["level0"], d["stage_seconds_rank0"]["conditional"])
PY
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -x -k "cfg4 or cfg3_full_size_headline_schedule_equals_oracle" > $O/pytest_b.txt 2>&1; tail -3 $O/pytest_b.txt
