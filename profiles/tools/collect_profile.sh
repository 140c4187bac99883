#!/bin/bash
# Profile of bench.py on one MI355X: kernel-trace stats + three separate PMC passes (never combined with tracing
# domains other than --kernel-trace).  usage: [ROUND=r03] bash profiles/tools/collect_profile.sh <config> [extra bench args]
# Writes gpurun_out/prof_r02_<config>/.  Every rocprofv3 call sits under `timeout` and writes CSV.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
CFG=$1; shift
ROUND=${ROUND:-r05}
OUT=$PWD/gpurun_out/prof_${ROUND}_$CFG
rm -rf "$OUT"; mkdir -p "$OUT"
BENCH="python $PWD/bench.py --config $CFG --no-cpu-baseline --no-other-schedule --no-one-chain $*"
ROOT=$PWD
cd /tmp
rm -rf /tmp/prof_stats /tmp/pmc_sq /tmp/pmc_f /tmp/pmc_w
timeout ${PROF_TIMEOUT:-900} rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- $BENCH --steps ${STATS_STEPS:-2} --warmup ${STATS_WARMUP:-1} > "$OUT/bench_under_rocprof.json" 2> /tmp/prof_stats.err
find /tmp/prof_stats -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats.csv" \;
timeout ${PROF_TIMEOUT:-900} rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES --output-format csv -d /tmp/pmc_sq -- $BENCH --steps 1 --warmup 0 > /dev/null 2>&1
timeout ${PROF_TIMEOUT:-900} rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d /tmp/pmc_f -- $BENCH --steps 1 --warmup 0 > /dev/null 2>&1
timeout ${PROF_TIMEOUT:-900} rocprofv3 --kernel-trace --pmc TCC_MISS_sum TCC_REQ_sum WRITE_SIZE --output-format csv -d /tmp/pmc_w -- $BENCH --steps 1 --warmup 0 > /dev/null 2>&1
# r05: the instruction mix of the kernel (two more passes, SQ counters only): Float64 add / mul / fma / transcendental, Float32, integer, conversions
rm -rf /tmp/pmc_m1 /tmp/pmc_m2
timeout ${PROF_TIMEOUT:-900} rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_BUSY_CYCLES --output-format csv -d /tmp/pmc_m1 -- $BENCH --steps 1 --warmup 0 > /dev/null 2>/tmp/pmc_m1.err
timeout ${PROF_TIMEOUT:-900} rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 --output-format csv -d /tmp/pmc_m2 -- $BENCH --steps 1 --warmup 0 > /dev/null 2>/tmp/pmc_m2.err
tail -3 /tmp/pmc_m1.err /tmp/pmc_m2.err > "$OUT/pmc_mix_err.txt" 2>/dev/null
for d in sq f w m1 m2; do python "$ROOT/profiles/tools/pmc_sum.py" /tmp/pmc_$d > "$OUT/pmc_$d.json"; done
ls -la "$OUT"
