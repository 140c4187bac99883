#!/bin/bash
# r04 run B: (1) parity of the screened size-3 test + unscaled division (fz tests, fuzz, cfg3 full size); (2) cfg3 bench + PMC;
# (3) attribution of the discrete persistent kernel: host trace + FW_MI_TICKS build on cfg4 (ff 1 / 0) and cfg2, PMC of the
# level-synchronous form (FW_MI_ROUNDS=1) as the "tests alone" reference
set -u
cd "${GRAFT_REPO_ROOT:-.}"; ROOT=$PWD; export TMPDIR=/tmp
O=$ROOT/gpurun_out/r4_b; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_fz.py tests/test_gpu_fuzz.py tests/test_gpu_fznz.py -q -x > $O/pytest_fz.txt 2>&1; tail -3 $O/pytest_fz.txt
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -x -k "cfg3" > $O/pytest_cfg3.txt 2>&1; tail -3 $O/pytest_cfg3.txt
python bench.py --steps 10 --warmup 1 --no-cpu-baseline > $O/bench_cfg3.json 2> $O/bench_cfg3.err
python - <<PY
import json
d=json.loads(open("$O/bench_cfg3.json").read().strip().splitlines()[-1])
print("cfg3 ms", d["ms_per_step"], "other", d["other_schedule"]["ms_per_step"], "frac", d["roofline"]["frac"], "edges", d["edges"], "eval", d["tests_per_step"]["conditional_evaluated"], "kernel_s", d["roofline"]["kernel_seconds_per_step"])
PY
ROUND=r04 bash profiles/tools/collect_profile.sh cfg3 > $O/collect_cfg3.log 2>&1
python profiles/tools/install_profile.py cfg3 fz_subsets_seg r04 > $O/install_cfg3.txt 2>&1; head -30 $O/install_cfg3.txt
# (3) discrete kernel
export FW_KNOBS=1
for ff in 1 0; do
  FW_TRACE_HOST=1 python bench.py --config cfg4 --feed-forward $ff --steps 1 --warmup 1 --no-cpu-baseline --no-other-schedule --no-one-chain > $O/cfg4_ff${ff}.json 2> $O/cfg4_ff${ff}_trace.txt
  FW_LIB_PATH=$ROOT/flashweave.jl_amd/libflashweave_amd_ticks.so FW_TRACE_HOST=1 python bench.py --config cfg4 --feed-forward $ff --steps 1 --warmup 1 --no-cpu-baseline --no-other-schedule --no-one-chain > $O/cfg4_ff${ff}_ticks.json 2> $O/cfg4_ff${ff}_ticks_trace.txt
done
FW_LIB_PATH=$ROOT/flashweave.jl_amd/libflashweave_amd_ticks.so FW_TRACE_HOST=1 python bench.py --config cfg2 --steps 1 --warmup 1 --no-cpu-baseline --no-other-schedule --no-one-chain > $O/cfg2_ticks.json 2> $O/cfg2_ticks_trace.txt
FW_MI_ROUNDS=1 ROUND=r04rounds bash profiles/tools/collect_profile.sh cfg4 --feed-forward 0 > $O/collect_cfg4_rounds.log 2>&1
cp -r gpurun_out/prof_r04rounds_cfg4 $O/ 2>/dev/null
grep -h 'state machine\|test routine\|boards \|team rounds' $O/cfg4_ff1_ticks_trace.txt | tail -12
