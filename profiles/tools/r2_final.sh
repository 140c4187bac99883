O=gpurun_out/r2_final; mkdir -p $O
bash profiles/tools/collect_r02.sh cfg3 > $O/collect_cfg3.log 2>&1
bash profiles/tools/collect_r02.sh cfg4 > $O/collect_cfg4.log 2>&1
timeout 900 python bench.py --steps 10 --warmup 1 --host-seam > $O/bench_cfg3_n1.json 2> $O/bench_cfg3_n1.err
timeout 600 python bench.py --config cfg4 --steps 5 --warmup 1 > $O/bench_cfg4_n1.json 2> $O/bench_cfg4_n1.err
timeout 600 python bench.py --config cfg2 --steps 10 --warmup 1 > $O/bench_cfg2_n1.json 2> $O/bench_cfg2_n1.err
timeout 600 python bench.py --gpus 1 --force-dist --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_cfg3_forcedist_nccl.json 2> $O/bench_cfg3_forcedist_nccl.err
timeout 600 python bench.py --gpus 2 --backend gloo --single-device --config cfg4 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_cfg4_2rank_gloo_1gpu.json 2> $O/bench_cfg4_2rank_gloo_1gpu.err
ls -la $O
