#!/bin/bash
# quick SQ-counter pass of cfg3 (one pass): VALU / SALU per test, busy and wait shares of the segment kernel
set -u
R=$PWD; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pmc_q /tmp/pmc_q2
BENCH="python $R/bench.py --config cfg3 --no-cpu-baseline --no-other-schedule --no-one-chain --steps 1 --warmup 0"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES --output-format csv -d /tmp/pmc_q -- $BENCH > /tmp/q.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INST_CYCLES_VMEM --output-format csv -d /tmp/pmc_q2 -- $BENCH > /dev/null 2>&1
mkdir -p $R/gpurun_out/r5_pmc_quick
python $R/profiles/tools/pmc_sum.py /tmp/pmc_q > $R/gpurun_out/r5_pmc_quick/sq.json
python $R/profiles/tools/pmc_sum.py /tmp/pmc_q2 > $R/gpurun_out/r5_pmc_quick/sq2.json
tail -1 /tmp/q.json > $R/gpurun_out/r5_pmc_quick/bench.json
python - <<PY
import json
b=json.loads(open("$R/gpurun_out/r5_pmc_quick/bench.json").read()); ev=b["tests_per_step"]["conditional_evaluated"]
for f in ("sq","sq2"):
    d=json.load(open("$R/gpurun_out/r5_pmc_quick/%s.json"%f))
    for k,v in d.items():
        if "fz_subsets_seg" in k:
            print(k[:60], {c:(x/ev if c.startswith("SQ_INSTS") else x) for c,x in v.items()})
            if "SQ_WAVE_CYCLES" in v: print("valu active frac", v["SQ_ACTIVE_INST_VALU"]/v["SQ_WAVE_CYCLES"], "wait any", v["SQ_WAIT_INST_ANY"]/v["SQ_WAVE_CYCLES"])
PY
