#!/bin/bash
# r04 final collection, part C (final kernels: no transposed gathers, 32-bit binomials in the table kernel's prologue): GPU test suite,
# bench lines of every config, kernel stats + PMC of cfg2 / cfg3 / cfg5, one-chain trace of cfg3
set -u
cd "${GRAFT_REPO_ROOT:-.}"; ROOT=$PWD; export TMPDIR=/tmp
O=$ROOT/gpurun_out/r4_final_c; rm -rf $O; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; grep -E 'passed|failed' $O/pytest_gpu.txt
ROUND=r04 bash profiles/tools/collect_profile.sh cfg2 > $O/collect_cfg2.log 2>&1
ROUND=r04 bash profiles/tools/collect_profile.sh cfg3 > $O/collect_cfg3.log 2>&1
cd /tmp; rm -rf /tmp/oc
FW_KNOBS=1 FW_DH_CHAINS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/oc -- python $ROOT/bench.py --no-cpu-baseline --no-other-schedule --no-one-chain --steps 3 --warmup 1 > $O/cfg3_one_chain_bench_under_rocprof.json 2> /dev/null
find /tmp/oc -name '*kernel_stats.csv' -exec cp {} $O/cfg3_one_chain_kernel_stats.csv \;
cd $ROOT
python bench.py --config cfg5 --steps 1 --warmup 1 --cpu-seconds 10 > $O/bench_cfg5_n1.json 2>/dev/null
ROUND=r04 bash profiles/tools/collect_profile.sh cfg5 > $O/collect_cfg5.log 2>&1
python bench.py --config cfg2 --steps 10 --warmup 2 > $O/bench_cfg2_n1.json 2>/dev/null
python bench.py --config cfg4 --steps 5 --warmup 1 > $O/bench_cfg4_n1.json 2>/dev/null
python bench.py --config cfg3he --steps 5 --warmup 1 > $O/bench_cfg3he_n1.json 2>/dev/null
python bench.py --steps 20 --warmup 1 --host-seam > $O/bench_cfg3_n1.json 2> $O/bench_cfg3_n1.err
python - <<PY
import json
for c in ("cfg3","cfg2","cfg4","cfg3he","cfg5"):
    d=json.loads(open("$O/bench_%s_n1.json"%c).read().strip().splitlines()[-1])
    print(c,"ms", round(d["ms_per_step"],2), "other", d["other_schedule"] and round(d["other_schedule"]["ms_per_step"],2), "edges", d["edges"], "frac", round(d["roofline"]["frac"],4), d["roofline"]["bound"], "value %.3g"%d["value"])
PY
