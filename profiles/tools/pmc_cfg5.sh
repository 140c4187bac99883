# one SQ counter pass over a full cfg5 pass (no tracing domains besides --kernel-trace): instruction counts of the size-4/5 kernels
set -u
cd "${GRAFT_REPO_ROOT:-.}"; ROOT=$PWD; export TMPDIR=/tmp
O=$ROOT/gpurun_out/r3_pmc5; mkdir -p $O
cd /tmp && rm -rf /tmp/pmc5
timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES --output-format csv -d /tmp/pmc5 -- python $ROOT/bench.py --config cfg5 --steps 1 --warmup 0 --no-cpu-baseline --no-other-schedule --no-one-chain > $O/bench_under_pmc.json 2> /dev/null
python $ROOT/profiles/tools/pmc_sum.py /tmp/pmc5 > $O/pmc_sq.json
python - <<PY
import json
d=json.load(open("$O/pmc_sq.json"))
for k,v in d.items():
    if "fz_subsets_seg" in k: print(k, {a: v[a] for a in v})
PY
