O=gpurun_out/s3g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fz.py tests/test_gpu_fuzz.py -x -q > $O/pytest.txt 2>&1; echo rc=$? >> $O/pytest.txt
tail -15 $O/pytest.txt
ABL_K=5 ABL_LONG=1 python profiles/ablate_fz.py 2>/dev/null
echo no-l1t; FW_FZ_DBG=1 ABL_K=5 ABL_LONG=1 python profiles/ablate_fz.py 2>/dev/null
