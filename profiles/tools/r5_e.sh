#!/bin/bash
# r05 run E: the rest of pytest -m gpu after the fz_nz arena fix (dist + fullsize passed in run D), the matrix / vector overlap probe
O=gpurun_out/r5_e; mkdir -p $O
timeout 300 profiles/tools/mfma_overlap_probe.bin > $O/mfma_overlap.txt 2>&1
cat $O/mfma_overlap.txt
timeout 1800 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_fz.py tests/test_gpu_fznz.py tests/test_gpu_fzs.py tests/test_gpu_mi.py tests/test_gpu_norm.py -q 2>&1 | tail -25 > $O/pytest.txt
cat $O/pytest.txt
