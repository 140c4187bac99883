#!/bin/bash
# r05: ablation builds of the level-0 matrix loop (epilogue off: FW_L0_DBG=1): no barrier / no staging pieces / both / compiler's own order.
# The variants are builds of fw_mi.o alone, linked against the other objects (the L0M_ABL knob lived in the kernel while this was measured;
# profiles/r05_level0_matrix_loop.json holds the numbers):
#   for v in ABL=1 ABL=2 ABL=3 SCHED=0; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DL0M_$v -c fw_mi.hip -o /tmp/fw_mi_$v.o;
#     hipcc --offload-arch=gfx950 -shared -fPIC -o ../libfw_${v/=/}.so <the other objects> /tmp/fw_mi_$v.o -ldl; done
O=gpurun_out/r5_level0_ablate; mkdir -p $O
for a in base ABL1 ABL2 ABL3 SCHED0; do
  lib=$PWD/flashweave.jl_amd/libfw_$a.so; [ $a = base ] && lib=$PWD/flashweave.jl_amd/libflashweave_amd.so
  echo "== $a" >> $O/l0_ablate.txt
  FW_LIB_PATH=$lib FW_KNOBS=1 L0_ABLATE_SET="1" timeout 600 python profiles/tools/l0_ablate.py 2>&1 | grep -E "^[0-9] " | tail -1 >> $O/l0_ablate.txt
done
cat $O/l0_ablate.txt
