#!/bin/bash
# r05: where the host graph passes at the end of fw_learn_network spend their time
export FW_KNOBS=1 FW_TRACE_HOST=1
O=gpurun_out/r5_host_trace; mkdir -p $O
nproc; python -c "import os; print(len(os.sched_getaffinity(0)), os.cpu_count())"; cat /sys/fs/cgroup/cpu.max 2>/dev/null
for cfg in cfg3 cfg4; do
  timeout 400 python bench.py --config $cfg --steps 6 --warmup 2 --no-cpu-baseline 2>$O/err_$cfg.txt | tail -1 | cut -c1-200
  grep "symmetric graph\|edges pass" $O/err_$cfg.txt | tail -6
done
timeout 600 python -m pytest tests/test_gpu_fz.py tests/test_gpu_fuzz.py tests/test_gpu_fznz.py -m gpu -q -x 2>&1 | tail -2
