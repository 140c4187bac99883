#!/bin/bash
# r04 run J: the self-service device round (FW_DH_FUSE=3: every target fills its own slice of the segment list) -- parity, timing against the
# three-launch round; cfg5 with and without the transposed gathers
set -u
cd "${GRAFT_REPO_ROOT:-.}"; ROOT=$PWD; export TMPDIR=/tmp
O=$ROOT/gpurun_out/r4_j; rm -rf $O; mkdir -p $O
export FW_KNOBS=1
timeout 900 python -m pytest tests/test_gpu_fz.py tests/test_gpu_fznz.py tests/test_gpu_dist.py -q -x > $O/pytest_fz.txt 2>&1; grep -E 'passed|failed|^E ' $O/pytest_fz.txt | head -5
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -q -x -k "cfg3 or cfg5_parameters" > $O/pytest_full.txt 2>&1; grep -E 'passed|failed|^E ' $O/pytest_full.txt | head -5
timeout 900 python -m tests.fuzz_gpu --first 240000 --cases 800 > $O/fuzz.txt 2>&1; tail -1 $O/fuzz.txt
FW_DH_SPEC=8 FW_DH_SPEC0=4 FW_DH_SPEC1=4 FW_DH_SPEC_BELOW=100000000000 FW_DH_SPEC0_BELOW=100000000000 FW_DH_SPEC0_JOBS=100000 FW_DH_CHAINS=2 FW_DH_CHAIN_MIN=4 FW_DH_BATCH=3 timeout 900 python -m tests.fuzz_gpu --first 250000 --cases 600 > $O/fuzz_forced.txt 2>&1; tail -1 $O/fuzz_forced.txt
for f in 0 3; do
  FW_DH_FUSE=$f python bench.py --steps 10 --warmup 1 --no-cpu-baseline > $O/cfg3_fuse$f.json 2>/dev/null
  FW_DH_FUSE=$f python bench.py --simulate-world 8 --simulate-rank -1 --steps 2 --warmup 1 --no-cpu-baseline --no-other-schedule --no-one-chain > $O/sim8_fuse$f.json 2>/dev/null
done
python - <<PY
import json
for f in (0,3):
    d=json.loads(open("$O/cfg3_fuse%d.json"%f).read().strip().splitlines()[-1])
    print("fuse",f,"ms", round(d["ms_per_step"],2), "other", round(d["other_schedule"]["ms_per_step"],2), "one-chain step", round(1e3*d["roofline"]["step_seconds_of_that_pass"],2), "edges", d["edges"], "launches/step", d["kernel_launches_per_step"], "eval", d["tests_per_step"]["conditional_evaluated"])
    d=json.loads(open("$O/sim8_fuse%d.json"%f).read().strip().splitlines()[-1])
    print("  sim8 slowest", round(d["ms_per_step"],2), [round(x,1) for x in d["simulated_world"]["ms_per_step_by_rank"]])
PY
for v in nocort default; do
  if [ $v = default ]; then unset FW_LIB_PATH; else export FW_LIB_PATH=$ROOT/flashweave.jl_amd/libflashweave_amd_$v.so; fi
  python bench.py --config cfg5 --steps 1 --warmup 0 --no-cpu-baseline --no-other-schedule --no-one-chain > $O/cfg5_$v.json 2>/dev/null
  python -c "
import json
d=json.loads(open('$O/cfg5_$v.json').read().strip().splitlines()[-1]); print('cfg5 $v s', round(d['ms_per_step']/1e3,2), 'edges', d['edges'])"
done
