#!/bin/bash
# r04 run H: binned_nz_clr beyond 16 384 samples (keys through device memory), variant S (thresholds + lazy p, unit sqrt / division, size-class
# gram kernel): parity and time; n > 65 535 through the persistent kernel
set -u
cd "${GRAFT_REPO_ROOT:-.}"; ROOT=$PWD; export TMPDIR=/tmp
O=$ROOT/gpurun_out/r4_h; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_norm.py tests/test_gpu_fzs.py -q -x > $O/pytest_a.txt 2>&1; grep -E 'passed|failed|^E ' $O/pytest_a.txt | head
timeout 900 python -m pytest tests/test_gpu_mi.py -q -x -k "65535" > $O/pytest_b.txt 2>&1; grep -E 'passed|failed|^E ' $O/pytest_b.txt | head
python profiles/tools/fzs_micro.py > $O/fzs_micro.txt 2>&1; tail -5 $O/fzs_micro.txt | cut -c1-300
python bench.py --stream-columns --steps 1 --warmup 1 --no-cpu-baseline --no-other-schedule > $O/bench_cfg3_stream_columns_whole.json 2>/dev/null
python - <<PY
import json
d=json.loads(open("$O/bench_cfg3_stream_columns_whole.json").read().strip().splitlines()[-1])
print("variant S whole ms", round(d["ms_per_step"],1), "edges", d["edges"], "roofline", {k:d["roofline"][k] for k in ("bound","achieved","frac")}, d["roofline"].get("job_matrices",{}).get("jobs"))
PY
timeout 1200 python -m tests.fuzz_gpu --first 140000 --cases 600 > $O/fuzz.txt 2>&1; tail -2 $O/fuzz.txt
