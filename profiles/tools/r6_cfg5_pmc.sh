#!/bin/bash
# r06: cfg5's counter passes again on the final defaults (8 192 segments per launch, 32-bit row base): SQ, FETCH, WRITE passes only
set -u
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/prof_r06_cfg5; mkdir -p $OUT
BENCH="python $ROOT/bench.py --config cfg5 --no-cpu-baseline --no-other-schedule --no-one-chain --steps 1 --warmup 0"
cd /tmp; rm -rf /tmp/pmc_sq /tmp/pmc_f /tmp/pmc_w
timeout 260 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES --output-format csv -d /tmp/pmc_sq -- $BENCH > /dev/null 2>&1
timeout 260 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d /tmp/pmc_f -- $BENCH > /dev/null 2>&1
timeout 260 rocprofv3 --kernel-trace --pmc TCC_MISS_sum TCC_REQ_sum WRITE_SIZE --output-format csv -d /tmp/pmc_w -- $BENCH > /dev/null 2>&1
for d in sq f w; do python "$ROOT/profiles/tools/pmc_sum.py" /tmp/pmc_$d > "$OUT/pmc_$d.json"; done
ls -la $OUT
