O=gpurun_out/s3d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fz.py tests/test_gpu_fuzz.py -x -q > $O/pytest.txt 2>&1; echo rc=$? >> $O/pytest.txt
tail -5 $O/pytest.txt
ABL_K=5 python profiles/ablate_fz.py 2>/dev/null
echo generic; FW_NO_HK=1 ABL_K=5 python profiles/ablate_fz.py 2>/dev/null
