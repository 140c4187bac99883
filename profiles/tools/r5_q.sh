#!/bin/bash
export FW_KNOBS=1
O=gpurun_out/r5_q; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_mi.py tests/test_gpu_maxk.py -q -x 2>&1 | tail -45 > $O/pytest_all.txt; grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" $O/pytest_all.txt | tail -40
