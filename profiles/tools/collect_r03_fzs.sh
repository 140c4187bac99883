#!/bin/bash
# streaming kernel (final form of the round): micro-benchmark, kernel stats, traffic counters in two separate passes
set -u
cd "${GRAFT_REPO_ROOT:-.}"; ROOT=$PWD; export TMPDIR=/tmp
O=$ROOT/gpurun_out/prof_r03_fzs; rm -rf $O; mkdir -p $O
python profiles/tools/fzs_micro.py 40 2000 > $O/micro.json 2>/dev/null
python profiles/tools/fzs_micro.py 100 200 >> $O/micro.json 2>/dev/null
cd /tmp; rm -rf /tmp/fzs_stats /tmp/fzs_f /tmp/fzs_w
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fzs_stats -- python $ROOT/profiles/tools/fzs_micro.py 40 2000 > $O/micro_under_rocprof.json 2>/dev/null
find /tmp/fzs_stats -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats.csv \;
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d /tmp/fzs_f -- python $ROOT/profiles/tools/fzs_micro.py 40 2000 > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCC_MISS_sum TCC_REQ_sum WRITE_SIZE --output-format csv -d /tmp/fzs_w -- python $ROOT/profiles/tools/fzs_micro.py 40 2000 > /dev/null 2>&1
python $ROOT/profiles/tools/pmc_sum.py /tmp/fzs_f > $O/pmc_f.json
python $ROOT/profiles/tools/pmc_sum.py /tmp/fzs_w > $O/pmc_w.json
cd $ROOT
python bench.py --stream-columns --max-targets 9800 --steps 2 --warmup 1 --no-other-schedule --no-cpu-baseline > $O/bench_stream.json 2>/dev/null
python -m pytest tests/test_gpu_fzs.py tests/test_gpu_fuzz.py -q 2>&1 | tail -2
cat $O/micro.json | cut -c1-400
