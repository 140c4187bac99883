#!/bin/bash
# r04 run E: instruction issue costs (valu_rate), saturated segment-kernel throughput r03 vs r04 (ablate_fz.py), and the persistent
# discrete kernel with its launch context / records in LDS: parity + timing + tick profile
set -u
cd "${GRAFT_REPO_ROOT:-.}"; ROOT=$PWD; export TMPDIR=/tmp
O=$ROOT/gpurun_out/r4_e; rm -rf $O; mkdir -p $O
./flashweave.jl_amd/valu_rate.bin > $O/valu_rate.txt 2>&1; cat $O/valu_rate.txt
export FW_KNOBS=1
echo "--- ablate r03" ; FW_LIB_PATH=$ROOT/flashweave.jl_amd/libflashweave_amd_r03.so python profiles/ablate_fz.py 2>/dev/null | tee $O/ablate_r03.txt
echo "--- ablate r04" ; python profiles/ablate_fz.py 2>/dev/null | tee $O/ablate_r04.txt
timeout 1200 python -m pytest tests/test_gpu_mi.py tests/test_gpu_fuzz.py -q -x > $O/pytest_mi.txt 2>&1; tail -3 $O/pytest_mi.txt
for cfg in cfg4 cfg2; do
  python bench.py --config $cfg --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_$cfg.json 2>/dev/null
  FW_LIB_PATH=$ROOT/flashweave.jl_amd/libflashweave_amd_r03.so python bench.py --config $cfg --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_${cfg}_r03lib.json 2>/dev/null
done
python - <<PY
import json
for c in ("cfg4","cfg4_r03lib","cfg2","cfg2_r03lib"):
    d=json.loads(open("$O/bench_%s.json"%c).read().strip().splitlines()[-1])
    print(c,"ms", round(d["ms_per_step"],2), "other", round(d["other_schedule"]["ms_per_step"],2), "edges", d["edges"], "eval", d["tests_per_step"]["conditional_evaluated"], "l0", round(1e3*d["stage_seconds_rank0"]["level0"],2), "cond", round(1e3*d["stage_seconds_rank0"]["conditional"],2))
PY
FW_LIB_PATH=$ROOT/flashweave.jl_amd/libflashweave_amd_ticks.so FW_TRACE_HOST=1 python bench.py --config cfg4 --feed-forward 0 --steps 1 --warmup 1 --no-cpu-baseline --no-other-schedule --no-one-chain > $O/cfg4_ff0_ticks.json 2> $O/cfg4_ff0_ticks_trace.txt
grep -h 'state machine\|test routine\|boards ' $O/cfg4_ff0_ticks_trace.txt | tail -3
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -x -k "cfg4" > $O/pytest_cfg4.txt 2>&1; tail -3 $O/pytest_cfg4.txt
