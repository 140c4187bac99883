#!/bin/bash
# r03 collection, part D (final state of the round): bench lines, cfg3 / cfg4 kernel stats + PMC, N-rank replay tables, fuzz, pytest log
set -u
cd "${GRAFT_REPO_ROOT:-.}"; ROOT=$PWD; export TMPDIR=/tmp
O=$ROOT/gpurun_out/r03_final_d; rm -rf $O; mkdir -p $O
python bench.py --steps 20 --warmup 1 --host-seam > $O/bench_cfg3_n1.json 2> $O/bench_cfg3_n1.err
python bench.py --config cfg2 --steps 10 --warmup 2 > $O/bench_cfg2_n1.json 2>/dev/null
python bench.py --config cfg4 --steps 5 --warmup 1 > $O/bench_cfg4_n1.json 2>/dev/null
python bench.py --config cfg3he --steps 5 --warmup 1 > $O/bench_cfg3he_n1.json 2>/dev/null
python bench.py --no-cor-matrix --max-targets 9800 --steps 2 --warmup 1 --no-other-schedule --no-cpu-baseline > $O/bench_cfg3_no_cor_matrix_first9800.json 2>/dev/null
python bench.py --gpus 1 --force-dist --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_cfg3_n1_nccl_world1.json 2> $O/nccl_world1.err
ROUND=r03 bash profiles/tools/collect_profile.sh cfg3 > $O/collect_cfg3.log 2>&1
ROUND=r03 bash profiles/tools/collect_profile.sh cfg4 > $O/collect_cfg4.log 2>&1
bash profiles/tools/simulate_world.sh cfg3 > $O/simulate_world_cfg3.txt 2>&1
cp gpurun_out/simulate_world/cfg3_n*.json $O/ 2>/dev/null
bash profiles/tools/simulate_world.sh cfg4 > $O/simulate_world_cfg4.txt 2>&1
cp gpurun_out/simulate_world/cfg4_n*.json $O/ 2>/dev/null
bash profiles/tools/r3_fuzz.sh > $O/fuzz.txt 2>&1
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1
tail -3 $O/pytest_gpu.txt; cat $O/fuzz.txt; cat $O/simulate_world_cfg3.txt $O/simulate_world_cfg4.txt
