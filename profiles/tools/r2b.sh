set -x
O=gpurun_out/r2b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mi.py tests/test_gpu_fuzz.py tests/test_gpu_dist.py tests/test_gpu_fullsize.py -x -q > $O/pytest.txt 2>&1; echo rc=$? >> $O/pytest.txt
tail -15 $O/pytest.txt
timeout 900 python bench.py --config cfg4 --steps 3 --warmup 1 --feed-forward 0 --no-other-schedule --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench_cfg4.err; echo rc=$?
timeout 600 python bench.py --config cfg2 --steps 5 --warmup 1 --feed-forward 0 --no-other-schedule --no-cpu-baseline > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo rc=$?
FW_DEV_MIN_TARGETS=64 timeout 600 python bench.py --config cfg2 --steps 5 --warmup 1 --feed-forward 0 --no-other-schedule --no-cpu-baseline > $O/bench_cfg2_dev.json 2> $O/bench_cfg2_dev.err; echo rc=$?
FW_MI_ROUNDS=1 timeout 900 python bench.py --config cfg4 --steps 3 --warmup 1 --feed-forward 0 --no-other-schedule --no-cpu-baseline > $O/bench_cfg4_rounds.json 2> $O/bench_cfg4_rounds.err; echo rc=$?
tail -3 $O/*.err
