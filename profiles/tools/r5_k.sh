#!/bin/bash
# r05 run K: kernel trace of cfg4 after the level-0 work + full parity of the discrete kinds (tests/test_gpu_mi.py, fuzz, cfg4 full size)
O=$PWD/gpurun_out/r5_k; mkdir -p $O
R=$PWD
timeout 1500 python -m pytest tests/test_gpu_mi.py tests/test_gpu_fuzz.py tests/test_gpu_dist.py -q -x 2>&1 | tail -4 > $O/pytest.txt
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -q -x -k "cfg4" 2>&1 | tail -3 >> $O/pytest.txt
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" $O/pytest.txt
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_stats
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python $R/bench.py --config cfg4 --no-cpu-baseline --no-other-schedule --no-one-chain --steps 3 --warmup 1 > $O/bench_under_rocprof.json 2> /tmp/prof_stats.err
find /tmp/prof_stats -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats.csv \;
grep -E "mi_level0|dh_mi_target|radix|bh_" $O/kernel_stats.csv | sed 's/(anonymous namespace):://g' | awk -F'",' '{print substr($1,1,70), $2,$3,$4}' | cut -c1-160
