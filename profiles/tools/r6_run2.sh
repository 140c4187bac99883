S=profiles/tools/r6_determinism.sh
mkdir -p gpurun_out/r6b
timeout 900 python -m pytest tests/test_gpu_determinism.py tests/test_gpu_mi.py -m gpu -x -q --durations=15 > gpurun_out/r6b/pytest_mi.txt 2>&1; tail -25 gpurun_out/r6b/pytest_mi.txt
$S fix_cfg4_ff1 1000 200 --config cfg4 --feed-forward 1
for c in cfg2 cfg4; do python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r6b/bench_$c.json 2> gpurun_out/r6b/bench_$c.err; tail -c 1500 gpurun_out/r6b/bench_$c.json; done
