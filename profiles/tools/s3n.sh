FW_TRACE_HOST=1 timeout 600 python -m pytest tests/test_gpu_fz.py -x -q -k "table_kernels_equal_gather" -s 2>&1 | grep -v amdgpu | grep "longest accepted\|passed\|failed\|Error\|assert" | tail -12
