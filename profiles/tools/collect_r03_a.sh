#!/bin/bash
# r03 collection, part A (Fisher-z): cfg3 kernel stats + PMC, cfg5 kernel stats + SQ counters, streaming kernel micro-benchmark
set -u
cd "${GRAFT_REPO_ROOT:-.}"; ROOT=$PWD; export TMPDIR=/tmp
ROUND=r03 bash profiles/tools/collect_profile.sh cfg3 > gpurun_out/collect_cfg3.log 2>&1
O=$ROOT/gpurun_out/prof_r03_fzs; rm -rf $O; mkdir -p $O
cd /tmp; rm -rf /tmp/fzs_stats /tmp/fzs_pmc
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fzs_stats -- python $ROOT/profiles/tools/fzs_micro.py 40 2000 > $O/micro_under_rocprof.json 2>/dev/null
find /tmp/fzs_stats -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats.csv \;
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_REQ_sum --output-format csv -d /tmp/fzs_pmc -- python $ROOT/profiles/tools/fzs_micro.py 40 2000 > /dev/null 2>&1
python $ROOT/profiles/tools/pmc_sum.py /tmp/fzs_pmc > $O/pmc_f.json
cd $ROOT
python profiles/tools/fzs_micro.py 40 2000 > $O/micro.json 2>/dev/null
python profiles/tools/fzs_micro.py 100 200 >> $O/micro.json 2>/dev/null
# cfg5: kernel stats of a full pass
cd /tmp; rm -rf /tmp/c5_stats
O5=$ROOT/gpurun_out/prof_r03_cfg5; rm -rf $O5; mkdir -p $O5
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c5_stats -- python $ROOT/bench.py --config cfg5 --steps 1 --warmup 0 --no-cpu-baseline --no-other-schedule --no-one-chain > $O5/bench_under_rocprof.json 2>/dev/null
find /tmp/c5_stats -name '*kernel_stats.csv' -exec cp {} $O5/kernel_stats.csv \;
cd $ROOT
bash profiles/tools/pmc_cfg5.sh > gpurun_out/collect_cfg5_pmc.log 2>&1
ls -la gpurun_out/prof_r03_cfg3 $O $O5 gpurun_out/r3_pmc5
