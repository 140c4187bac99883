#!/bin/bash
export FW_KNOBS=1
bash profiles/tools/r6_ab2.sh
timeout 900 python -m pytest tests/test_gpu_fz.py tests/test_gpu_fuzz.py tests/test_gpu_dist.py -m gpu -q -x 2>&1 | tail -3
