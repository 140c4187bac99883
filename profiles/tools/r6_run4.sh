mkdir -p gpurun_out/r6d
bash profiles/tools/r6_ab.sh cfg3 2 libfw_prev.so libfw_r64.so libflashweave_amd.so
timeout 1200 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_fz.py tests/test_gpu_fznz.py tests/test_gpu_fuzz.py -m gpu -q --durations=25 -x > gpurun_out/r6d/pytest.txt 2>&1; tail -35 gpurun_out/r6d/pytest.txt
