#!/bin/bash
# r06: counters of the cfg3 segment kernel on the screened / local-matrix build (the standard passes + LDS / L1 / wait counters)
export FW_KNOBS=1
ROUND=r06 bash profiles/tools/collect_profile.sh cfg3 > gpurun_out/collect_cfg3.log 2>&1
ls gpurun_out/prof_r06_cfg3
cd /tmp; export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --config cfg3 --no-cpu-baseline --no-other-schedule --no-one-chain --steps 1 --warmup 0"
rm -rf /tmp/pmc_x1 /tmp/pmc_x2 /tmp/pmc_x3
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/pmc_x1 -- $B > /dev/null 2>/tmp/pmc_x1.err
timeout 600 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr TA_TA_BUSY_sum --output-format csv -d /tmp/pmc_x2 -- $B > /dev/null 2>/tmp/pmc_x2.err
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d /tmp/pmc_x3 -- $B > /dev/null 2>/tmp/pmc_x3.err
for d in x1 x2 x3; do tail -2 /tmp/pmc_$d.err; python $GRAFT_REPO_ROOT/profiles/tools/pmc_sum.py /tmp/pmc_$d > $GRAFT_REPO_ROOT/gpurun_out/prof_r06_cfg3/pmc_$d.json; done
python - <<'PY'
import json,os
R=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/prof_r06_cfg3/"
for d in ("x1","x2","x3","sq"):
    try:
        j=json.load(open(R+"pmc_%s.json"%d))
        for k,v in j.items():
            if "fz_subsets_seg_kernel<false, false, true>" in k: print(d, json.dumps(v))
    except Exception as e: print(d,"ERR",e)
PY
