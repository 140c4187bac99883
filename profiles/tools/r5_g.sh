#!/bin/bash
# r05 run G: ablation builds of mi_level0_mfma_kernel (L0M_ABL 1 = no staging after the first stage / no barrier, 2 = operand words from
# registers instead of LDS, 3 = both), epilogue off (FW_L0_DBG=1): level-0 stage seconds at cfg4
O=gpurun_out/r5_g; mkdir -p $O
for a in 0 1 2 3; do
  lib=$PWD/flashweave.jl_amd/libfw_abl$a.so; [ $a = 0 ] && lib=$PWD/flashweave.jl_amd/libflashweave_amd.so
  echo "== L0M_ABL=$a" >> $O/l0_ablate.txt
  FW_LIB_PATH=$lib FW_KNOBS=1 L0_ABLATE_SET="1" timeout 600 python profiles/tools/l0_ablate.py 2>&1 | grep -E "^[0-9] " | tail -1 >> $O/l0_ablate.txt
done
cat $O/l0_ablate.txt
