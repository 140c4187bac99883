mkdir -p gpurun_out/r6p
timeout 1500 python -m pytest tests -m gpu -q --durations=60 > gpurun_out/r6p/pytest_gpu.txt 2>&1; tail -75 gpurun_out/r6p/pytest_gpu.txt
