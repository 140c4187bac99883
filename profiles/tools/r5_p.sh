#!/bin/bash
export FW_KNOBS=1
# r05 run P: conditioning sets of 6 and 7 variables (tests/test_gpu_maxk.py), the three fuzz seeds of the R = 1 device schedule, fz / mi regressions
O=gpurun_out/r5_p; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_maxk.py -q 2>&1 | tail -40 > $O/pytest_maxk.txt; grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" $O/pytest_maxk.txt | tail -25
