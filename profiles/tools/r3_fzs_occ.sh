#!/bin/bash
# streaming kernel: workgroups per CU the segment kernel is compiled for x tests finished together (rebuilds fw_fzs.o on the GPU box)
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_fzs.py tests/test_gpu_fuzz.py -q 2>&1 | tail -2
for cfg in "4 4" "4 2" "4 1" "3 4" "5 4"; do
  set -- $cfg
  touch flashweave.jl_amd/csrc/fw_fzs.hip
  make -C flashweave.jl_amd/csrc EXTRA="-DFZS_SEG_OCC=$1 -DFZS_GMAX=$2" > /dev/null 2>&1
  echo "FZS_SEG_OCC=$1 FZS_GMAX=$2"
  for sh in "40 2000" "100 200"; do python profiles/tools/fzs_micro.py $sh 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('  ', d['accepted'], round(d['tests_per_s_in_kernel']/1e8,3),'e8 tests/s', round(d['frac_of_8TBps'],3))"; done
done
touch flashweave.jl_amd/csrc/fw_fzs.hip; make -C flashweave.jl_amd/csrc > /dev/null 2>&1
