# final collection of round 2 (session 3): tests, smoke, profiles, bench lines.  Output: gpurun_out/s3_final/
O=gpurun_out/s3_final; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo rc=$? >> $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo rc=$? >> $O/smoke.txt
bash profiles/tools/collect_r02.sh cfg3 > $O/collect_cfg3.log 2>&1
bash profiles/tools/collect_r02.sh cfg4 > $O/collect_cfg4.log 2>&1
timeout 900 python bench.py --steps 10 --warmup 1 --host-seam > $O/bench_cfg3_n1.json 2> $O/bench_cfg3_n1.err
timeout 600 python bench.py --config cfg4 --steps 5 --warmup 1 > $O/bench_cfg4_n1.json 2> $O/bench_cfg4_n1.err
timeout 600 python bench.py --config cfg2 --steps 10 --warmup 1 > $O/bench_cfg2_n1.json 2> $O/bench_cfg2_n1.err
timeout 600 python bench.py --config cfg3he --steps 10 --warmup 1 > $O/bench_cfg3he_n1.json 2> $O/bench_cfg3he_n1.err
timeout 600 python bench.py --gpus 1 --force-dist --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_cfg3_forcedist_nccl.json 2> $O/bench_cfg3_forcedist_nccl.err
timeout 600 python bench.py --gpus 2 --backend gloo --single-device --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_cfg3_2rank_gloo_1gpu.json 2> $O/bench_cfg3_2rank_gloo_1gpu.err
FW_TRACE_HOST=1 timeout 1200 python bench.py --config cfg5 --steps 1 --warmup 0 --no-other-schedule --cpu-seconds 10 > $O/bench_cfg5_n1.json 2> $O/bench_cfg5_n1.err
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ks_cfg5
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_cfg5 -- python $GRAFT_REPO_ROOT/bench.py --config cfg5 --steps 1 --warmup 0 --no-cpu-baseline --no-other-schedule > $GRAFT_REPO_ROOT/$O/cfg5_bench_under_rocprof.json 2>/dev/null
find /tmp/ks_cfg5 -name '*kernel_stats.csv' -exec cp {} $GRAFT_REPO_ROOT/$O/cfg5_kernel_stats.csv \;
cd $GRAFT_REPO_ROOT
tail -3 $O/pytest_gpu.txt; cat $O/smoke.txt | tail -2; ls -la $O
