mkdir -p gpurun_out/r6f
bash profiles/tools/r6_occ2.sh
timeout 1500 python -m pytest tests -m gpu -q --durations=15 > gpurun_out/r6f/pytest_gpu.txt 2>&1; tail -25 gpurun_out/r6f/pytest_gpu.txt
