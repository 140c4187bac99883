#!/bin/bash
# r04 final collection, part A: GPU test suite, bench lines (cfg3 headline with CPU leg and host seam, cfg2, cfg4, cfg3he), kernel stats + PMC of
# cfg3 (two chains and one chain) and cfg4, N-rank replay tables
set -u
cd "${GRAFT_REPO_ROOT:-.}"; ROOT=$PWD; export TMPDIR=/tmp
O=$ROOT/gpurun_out/r4_final_a; rm -rf $O; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
python bench.py --steps 20 --warmup 1 --host-seam > $O/bench_cfg3_n1.json 2> $O/bench_cfg3_n1.err
python bench.py --config cfg2 --steps 10 --warmup 2 > $O/bench_cfg2_n1.json 2>/dev/null
python bench.py --config cfg4 --steps 5 --warmup 1 > $O/bench_cfg4_n1.json 2>/dev/null
python bench.py --config cfg3he --steps 5 --warmup 1 > $O/bench_cfg3he_n1.json 2>/dev/null
ROUND=r04 bash profiles/tools/collect_profile.sh cfg3 > $O/collect_cfg3.log 2>&1
ROUND=r04 bash profiles/tools/collect_profile.sh cfg4 > $O/collect_cfg4.log 2>&1
cd /tmp; rm -rf /tmp/oc
FW_KNOBS=1 FW_DH_CHAINS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/oc -- python $ROOT/bench.py --no-cpu-baseline --no-other-schedule --no-one-chain --steps 3 --warmup 1 > $O/cfg3_one_chain_bench_under_rocprof.json 2> /dev/null
find /tmp/oc -name '*kernel_stats.csv' -exec cp {} $O/cfg3_one_chain_kernel_stats.csv \;
cd $ROOT
bash profiles/tools/simulate_world.sh cfg3 > $O/simulate_world_cfg3.txt 2>&1
cp gpurun_out/simulate_world/cfg3_n*.json $O/ 2>/dev/null
bash profiles/tools/simulate_world.sh cfg4 > $O/simulate_world_cfg4.txt 2>&1
cp gpurun_out/simulate_world/cfg4_n*.json $O/ 2>/dev/null
cat $O/simulate_world_cfg3.txt $O/simulate_world_cfg4.txt
python - <<PY
import json
for c in ("cfg3","cfg2","cfg4","cfg3he"):
    d=json.loads(open("$O/bench_%s_n1.json"%c).read().strip().splitlines()[-1])
    print(c,"ms", round(d["ms_per_step"],2), "other", round(d["other_schedule"]["ms_per_step"],2), "edges", d["edges"], "frac", round(d["roofline"]["frac"],4), d["roofline"]["bound"], "value %.3g"%d["value"])
PY
