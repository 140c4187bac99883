#!/bin/bash
# r03 collection, part B: bench lines of every config, cfg4 kernel stats + PMC, streaming-kernel PMC, simulated N-rank tables, pytest log
set -u
cd "${GRAFT_REPO_ROOT:-.}"; ROOT=$PWD; export TMPDIR=/tmp
O=$ROOT/gpurun_out/r03_final; rm -rf $O; mkdir -p $O
python bench.py --steps 10 --warmup 1 --host-seam > $O/bench_cfg3_n1.json 2> $O/bench_cfg3_n1.err
python bench.py --config cfg2 --steps 10 --warmup 2 > $O/bench_cfg2_n1.json 2>/dev/null
python bench.py --config cfg4 --steps 5 --warmup 1 > $O/bench_cfg4_n1.json 2>/dev/null
python bench.py --config cfg3he --steps 5 --warmup 1 > $O/bench_cfg3he_n1.json 2>/dev/null
python bench.py --config cfg5 --steps 1 --warmup 0 --cpu-seconds 10 > $O/bench_cfg5_n1.json 2>/dev/null
python bench.py --stream-columns --max-targets 9800 --steps 2 --warmup 1 --no-other-schedule --no-cpu-baseline > $O/bench_cfg3_stream_first9800.json 2>/dev/null
python bench.py --gpus 1 --force-dist --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_cfg3_n1_nccl_world1.json 2> $O/nccl_world1.err
python bench.py --config cfg4 --gpus 1 --force-dist --steps 3 --warmup 1 --no-cpu-baseline --no-other-schedule > $O/bench_cfg4_n1_nccl_world1.json 2>> $O/nccl_world1.err
ROUND=r03 bash profiles/tools/collect_profile.sh cfg4 > $O/collect_cfg4.log 2>&1
# streaming kernel: traffic counters in two separate passes
cd /tmp; rm -rf /tmp/fzs_f /tmp/fzs_w
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d /tmp/fzs_f -- python $ROOT/profiles/tools/fzs_micro.py 40 2000 > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCC_MISS_sum TCC_REQ_sum WRITE_SIZE --output-format csv -d /tmp/fzs_w -- python $ROOT/profiles/tools/fzs_micro.py 40 2000 > /dev/null 2>&1
python $ROOT/profiles/tools/pmc_sum.py /tmp/fzs_f > $O/fzs_pmc_f.json
python $ROOT/profiles/tools/pmc_sum.py /tmp/fzs_w > $O/fzs_pmc_w.json
cd $ROOT
bash profiles/tools/simulate_world.sh cfg3 > $O/simulate_world_cfg3.txt 2>&1
bash profiles/tools/simulate_world.sh cfg4 > $O/simulate_world_cfg4.txt 2>&1
bash profiles/tools/simulate_world.sh cfg5 --max-targets 40000 --no-other-schedule > $O/simulate_world_cfg5_first40000.txt 2>&1
cp gpurun_out/simulate_world/*.json $O/ 2>/dev/null
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1
tail -3 $O/pytest_gpu.txt
ls $O
