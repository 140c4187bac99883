#!/bin/bash
# r04 run D (analysis): per-launch durations of the cfg3 one-chain pass (FW_DH_LOG), saturated throughput of the segment kernel on
# uniform batches (ablate_fz.py), parts of the discrete level-0 kernel (l0_ablate.py), cfg4 kernel stats
set -u
cd "${GRAFT_REPO_ROOT:-.}"; ROOT=$PWD; export TMPDIR=/tmp
O=$ROOT/gpurun_out/r4_d; rm -rf $O; mkdir -p $O
export FW_KNOBS=1
FW_DH_CHAINS=1 FW_DH_TIME_EVERY=1 FW_DH_LOG=$O/dh_log_one_chain.txt python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-schedule --no-one-chain > $O/cfg3_log_bench.json 2>/dev/null
FW_DH_TIME_EVERY=1 FW_DH_LOG=$O/dh_log_two_chains.txt python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-schedule --no-one-chain > $O/cfg3_log2_bench.json 2>/dev/null
python profiles/ablate_fz.py > $O/ablate_fz.txt 2>&1; cat $O/ablate_fz.txt
FW_L0_VERBOSE=1 L0_ABLATE_SET="0 1 4 5" python profiles/tools/l0_ablate.py > $O/l0_ablate.txt 2>&1; grep -v amdgpu.ids $O/l0_ablate.txt
cd /tmp; rm -rf /tmp/c4
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c4 -- python $ROOT/bench.py --config cfg4 --steps 2 --warmup 1 --no-cpu-baseline --no-other-schedule --no-one-chain > $O/cfg4_bench_under_rocprof.json 2>/dev/null
find /tmp/c4 -name '*kernel_stats.csv' -exec cp {} $O/cfg4_kernel_stats.csv \;
head -8 $O/cfg4_kernel_stats.csv | cut -c1-160
