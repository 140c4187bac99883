#!/bin/bash
# r04 run I: variant S A/B (r03 library, gram kernel in one size class, eager p-values, final), binned test, matrix perturbation
set -u
cd "${GRAFT_REPO_ROOT:-.}"; ROOT=$PWD; export TMPDIR=/tmp
O=$ROOT/gpurun_out/r4_i; rm -rf $O; mkdir -p $O
export FW_KNOBS=1
for v in r03 GRAM_ONE EAGER_P final; do
  if [ $v = final ]; then unset FW_LIB_PATH; else export FW_LIB_PATH=$ROOT/flashweave.jl_amd/libflashweave_amd_$v.so; fi
  for a in 6 40 100; do python profiles/tools/fzs_micro.py $a > $O/fzs_${v}_$a.txt 2>/dev/null; python - <<PY
import json
try:
    d=json.loads(open("$O/fzs_${v}_$a.txt").read().strip().splitlines()[-1]); print("$v", "a=$a", "%.3g tests/s in kernel"%d["tests_per_s_in_kernel"], "wall", round(d["wall_s"],4), "kernel", round(d["kernel_s"],4))
except Exception as e: print("$v $a failed", e)
PY
  done
  python bench.py --stream-columns --max-targets 9800 --steps 1 --warmup 1 --no-cpu-baseline --no-other-schedule > $O/sc_$v.json 2>/dev/null
  python -c "
import json
d=json.loads(open('$O/sc_$v.json').read().strip().splitlines()[-1]); print('$v first 9800 targets ms', round(d['ms_per_step'],1), 'edges', d['edges'], 'kernel s', round(d['stage_seconds_rank0']['subsets_kernels_device'],3))"
done
unset FW_LIB_PATH
timeout 600 python -m pytest tests/test_gpu_norm.py -q -x > $O/pytest_norm.txt 2>&1; grep -E 'passed|failed|^E ' $O/pytest_norm.txt | head -5
python profiles/tools/cor_perturb.py > $O/cor_perturb.json 2>/dev/null; cat $O/cor_perturb.json
