R=${GRAFT_REPO_ROOT:-$PWD}; cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ks_he
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_he -- python $R/bench.py --config cfg3he --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/cfg3he_under_rocprof.json 2>/dev/null
find /tmp/ks_he -name '*kernel_stats.csv' -exec cp {} $R/gpurun_out/cfg3he_kernel_stats.csv \;
head -6 $R/gpurun_out/cfg3he_kernel_stats.csv | cut -c1-160
