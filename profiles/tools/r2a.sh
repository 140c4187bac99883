set -x
O=gpurun_out/r2a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo rc=$? >> $O/pytest.txt
timeout 900 python bench.py --steps 5 --warmup 1 --host-seam > $O/bench_default.json 2> $O/bench_default.err; echo rc=$?
for R in 256 512 2048 4096; do timeout 300 python bench.py --steps 3 --warmup 1 --round-size $R --no-other-schedule --no-cpu-baseline > $O/bench_R$R.json 2> $O/bench_R$R.err; done
timeout 300 python bench.py --gpus 1 --force-dist --steps 3 --no-cpu-baseline > $O/force_dist.json 2> $O/force_dist.err; echo rc=$?
timeout 600 python bench.py --gpus 2 --backend gloo --single-device --steps 3 --no-cpu-baseline > $O/two_rank_gloo.json 2> $O/two_rank_gloo.err; echo rc=$?
tail -3 $O/pytest.txt
