#!/bin/bash
# r05 run F: matrix / vector overlap probe with dense random operands (power), and the level-0 matrix-core kernel at cfg4 with zeroed
# loads (FW_L0_DBG=2: same instruction stream on all-zero planes) / without the epilogue (1) -- phase cycles per tile and stage seconds
O=gpurun_out/r5_f; mkdir -p $O
timeout 300 profiles/tools/mfma_overlap_probe.bin > $O/mfma_overlap.txt 2>&1
grep -E "rnd|k_mfma_only |k_v5_dep |k_v5_dep_lds " $O/mfma_overlap.txt | cut -c1-150
for d in 0 2 1 3; do
  echo "== FW_L0_DBG=$d" >> $O/l0_ablate.txt
  FW_KNOBS=1 FW_L0_VERBOSE=1 L0_ABLATE_SET="$d" timeout 600 python profiles/tools/l0_ablate.py 2>&1 | grep -E "shader cycles|^[0-9] " | tail -3 >> $O/l0_ablate.txt
done
cat $O/l0_ablate.txt
