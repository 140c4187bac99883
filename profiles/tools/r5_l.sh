#!/bin/bash
# r05 run L: is dh_mi_target_kernel bound by instruction fetch?  I-cache counters of the cfg4 pass (own PMC passes, kernel trace only)
O=$PWD/gpurun_out/r5_l; mkdir -p $O
R=$PWD
cd /tmp; export TMPDIR=/tmp
BENCH="python $R/bench.py --config cfg4 --no-cpu-baseline --no-other-schedule --no-one-chain --steps 1 --warmup 0"
rm -rf /tmp/pmc_a /tmp/pmc_b /tmp/pmc_c
timeout 600 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d /tmp/pmc_a -- $BENCH > /dev/null 2>/tmp/pmc_a.err
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d /tmp/pmc_b -- $BENCH > /dev/null 2>/tmp/pmc_b.err
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH --output-format csv -d /tmp/pmc_c -- $BENCH > /dev/null 2>/tmp/pmc_c.err
tail -2 /tmp/pmc_a.err /tmp/pmc_b.err /tmp/pmc_c.err > $O/err.txt
for d in a b c; do python $R/profiles/tools/pmc_sum.py /tmp/pmc_$d > $O/pmc_$d.json; done
python - <<'PY'
import json
for d in "abc":
    j=json.load(open("/root/repo/gpurun_out/r5_l/pmc_%s.json"%d))
    for k,v in j.items():
        if "dh_mi_target" in k or "mi_level0_mfma" in k: print(d, k[:40], json.dumps(v))
PY
