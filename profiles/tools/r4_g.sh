#!/bin/bash
# r04 run G: library-side RCCL (world of one rank), persistent-kernel micro changes (conditional clocks, one 16-byte poll): parity + time
set -u
cd "${GRAFT_REPO_ROOT:-.}"; ROOT=$PWD; export TMPDIR=/tmp
O=$ROOT/gpurun_out/r4_g; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dist.py -q -x > $O/pytest_dist.txt 2>&1; tail -15 $O/pytest_dist.txt
timeout 900 python -m pytest tests/test_gpu_mi.py tests/test_gpu_fuzz.py -q -x > $O/pytest_mi.txt 2>&1; tail -2 $O/pytest_mi.txt
for cfg in cfg4 cfg2; do python bench.py --config $cfg --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_$cfg.json 2>/dev/null; done
python bench.py --gpus 1 --force-dist --library-rccl --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_cfg3_librccl_world1.json 2> $O/librccl.err
python bench.py --config cfg4 --gpus 1 --force-dist --library-rccl --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_cfg4_librccl_world1.json 2> $O/librccl4.err
python - <<PY
import json
for c in ("cfg4","cfg2","cfg3_librccl_world1","cfg4_librccl_world1"):
    try:
        d=json.loads(open("$O/bench_%s.json"%c).read().strip().splitlines()[-1])
        print(c,"ms", round(d["ms_per_step"],2), "other", d["other_schedule"] and round(d["other_schedule"]["ms_per_step"],2), "edges", d["edges"], "l0", round(1e3*d["stage_seconds_rank0"]["level0"],2), "cond", round(1e3*d["stage_seconds_rank0"]["conditional"],2), "exchange", d["exchange"])
    except Exception as e: print(c, "FAILED", e)
PY
tail -5 $O/librccl.err
