#!/bin/bash
# r05 run I: ablation builds of the new level-0 matrix loop (epilogue off): no barrier / no staging pieces / both / compiler's own order
O=gpurun_out/r5_i; mkdir -p $O
for a in base ABL1 ABL2 ABL3 SCHED0; do
  lib=$PWD/flashweave.jl_amd/libfw_$a.so; [ $a = base ] && lib=$PWD/flashweave.jl_amd/libflashweave_amd.so
  echo "== $a" >> $O/l0_ablate.txt
  FW_LIB_PATH=$lib FW_KNOBS=1 L0_ABLATE_SET="1" timeout 600 python profiles/tools/l0_ablate.py 2>&1 | grep -E "^[0-9] " | tail -1 >> $O/l0_ablate.txt
done
cat $O/l0_ablate.txt
