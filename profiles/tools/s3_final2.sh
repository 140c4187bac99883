# re-collection of the lines that depend on the number of concurrent chains (default back to 2)
O=gpurun_out/s3_final; mkdir -p $O
bash profiles/tools/collect_r02.sh cfg3 > $O/collect_cfg3.log 2>&1
timeout 900 python bench.py --steps 10 --warmup 1 --host-seam > $O/bench_cfg3_n1.json 2> $O/bench_cfg3_n1.err
timeout 600 python bench.py --gpus 1 --force-dist --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_cfg3_forcedist_nccl.json 2> $O/bench_cfg3_forcedist_nccl.err
timeout 600 python bench.py --gpus 2 --backend gloo --single-device --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_cfg3_2rank_gloo_1gpu.json 2> $O/bench_cfg3_2rank_gloo_1gpu.err
FW_DH_CHAINS=1 timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_cfg3_onechain.json 2> /dev/null
FW_TRACE_HOST=1 timeout 1200 python bench.py --config cfg5 --steps 1 --warmup 0 --no-other-schedule --cpu-seconds 10 > $O/bench_cfg5_n1.json 2> $O/bench_cfg5_n1.err
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ks_cfg5
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_cfg5 -- python $GRAFT_REPO_ROOT/bench.py --config cfg5 --steps 1 --warmup 0 --no-cpu-baseline --no-other-schedule > $GRAFT_REPO_ROOT/$O/cfg5_bench_under_rocprof.json 2>/dev/null
find /tmp/ks_cfg5 -name '*kernel_stats.csv' -exec cp {} $GRAFT_REPO_ROOT/$O/cfg5_kernel_stats.csv \;
cd $GRAFT_REPO_ROOT; ls -la $O | head -30
