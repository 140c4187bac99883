#!/bin/bash
# r03 collection, part C (after the team targets of the discrete kernel and the job-local matrices of the recursive_pcor = 0 path):
# bench lines of the configs they touch, cfg4 kernel stats + PMC, the variant-S micro-benchmark in both forms, simulated N-rank tables
# of cfg4, pytest log
set -u
cd "${GRAFT_REPO_ROOT:-.}"; ROOT=$PWD; export TMPDIR=/tmp
O=$ROOT/gpurun_out/r03_final_c; rm -rf $O; mkdir -p $O
python bench.py --steps 10 --warmup 1 --host-seam > $O/bench_cfg3_n1.json 2> $O/bench_cfg3_n1.err
python bench.py --config cfg2 --steps 10 --warmup 2 > $O/bench_cfg2_n1.json 2>/dev/null
python bench.py --config cfg4 --steps 5 --warmup 1 > $O/bench_cfg4_n1.json 2>/dev/null
python bench.py --stream-columns --max-targets 9800 --steps 2 --warmup 1 --no-other-schedule --no-cpu-baseline > $O/bench_cfg3_stream_first9800.json 2>/dev/null
python bench.py --stream-columns --steps 2 --warmup 1 --no-other-schedule --no-cpu-baseline > $O/bench_cfg3_stream_whole.json 2>/dev/null
export FW_KNOBS=1
FW_FZS_GRAM=0 python bench.py --stream-columns --max-targets 9800 --steps 2 --warmup 1 --no-other-schedule --no-cpu-baseline > $O/bench_cfg3_stream_first9800_per_test_streaming.json 2>/dev/null
for g in 1 0; do for sh in "40 2000" "100 200"; do FW_FZS_GRAM=$g python profiles/tools/fzs_micro.py $sh 2>/dev/null | tail -1 >> $O/fzs_micro_gram$g.json; done; done
unset FW_KNOBS
cd /tmp; rm -rf /tmp/fzs_stats
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fzs_stats -- python $ROOT/profiles/tools/fzs_micro.py 40 2000 > /dev/null 2>&1
find /tmp/fzs_stats -name '*kernel_stats.csv' -exec cp {} $O/fzs_micro_gram_kernel_stats.csv \;
cd $ROOT
ROUND=r03 bash profiles/tools/collect_profile.sh cfg4 > $O/collect_cfg4.log 2>&1
bash profiles/tools/simulate_world.sh cfg4 > $O/simulate_world_cfg4.txt 2>&1
cp gpurun_out/simulate_world/cfg4_n*.json $O/ 2>/dev/null
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1
tail -3 $O/pytest_gpu.txt
cat $O/simulate_world_cfg4.txt
ls $O
