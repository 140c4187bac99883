mkdir -p gpurun_out/r6e
python -c "import os; print('cpus', os.cpu_count())"
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -m gpu -q --durations=20 -x > gpurun_out/r6e/pytest.txt 2>&1; tail -30 gpurun_out/r6e/pytest.txt
