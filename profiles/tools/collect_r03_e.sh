#!/bin/bash
# r03 collection, part E: pytest log and variant-S bench lines of the final state
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r03_final_e; rm -rf $O; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
python bench.py --stream-columns --steps 2 --warmup 1 --no-other-schedule --no-cpu-baseline > $O/bench_cfg3_stream_whole.json 2>/dev/null
python bench.py --stream-columns --max-targets 9800 --steps 2 --warmup 1 --no-other-schedule --no-cpu-baseline > $O/bench_cfg3_stream_first9800.json 2>/dev/null
python bench.py --no-cor-matrix --max-targets 9800 --steps 2 --warmup 1 --no-other-schedule --no-cpu-baseline > $O/bench_cfg3_no_cor_matrix_first9800.json 2>/dev/null
python bench.py --steps 20 --warmup 1 --host-seam > $O/bench_cfg3_n1.json 2>/dev/null
tail -c 200 $O/bench_cfg3_n1.json
