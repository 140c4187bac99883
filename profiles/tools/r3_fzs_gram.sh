#!/bin/bash
# recursive_pcor = 0: job-local correlation matrices (default) against the streamed form (FW_FZS_GRAM=0)
cd $GRAFT_REPO_ROOT
export FW_KNOBS=1
timeout 900 python -m pytest tests/test_gpu_fzs.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -3
for g in 1 0; do
  echo "FW_FZS_GRAM=$g"
  for sh in "40 2000" "100 200"; do FW_FZS_GRAM=$g python profiles/tools/fzs_micro.py $sh 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('  ', d['accepted'], round(d['tests_per_s_in_kernel']/1e8,3),'e8 tests/s in the kernels', round(d['kernel_s']*1e3,2), 'ms; wall', round(d['wall_s']*1e3,2), 'ms')"; done
  FW_FZS_GRAM=$g python bench.py --stream-columns --max-targets 9800 --steps 2 --warmup 1 --no-other-schedule --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   first 9800 targets:', round(d['ms_per_step'],1), 'ms', d['edges'], 'edges; kernel s', round(d['roofline']['kernel_seconds_per_step'],3))"
done
python bench.py --stream-columns --steps 2 --warmup 1 --no-other-schedule --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   whole cfg3, recursive_pcor = 0:', round(d['ms_per_step'],1), 'ms', d['edges'], 'edges; value', d['value'])"
