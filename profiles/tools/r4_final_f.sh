#!/bin/bash
# r04 final collection F: first window of 4 096 ranks for fz jobs at max_k <= 3 -- parity (fz tests, schedule independence at full size, fuzz),
# cfg3 profile set (kernel trace + PMC, one-chain kernel trace) and headline bench line with the CPU leg
O=gpurun_out/r4_final_f; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_fz.py tests/test_gpu_fuzz.py tests/test_gpu_dist.py -q 2>&1 | grep -i "passed\|failed\|error" | tail -3 > $O/pytest.txt
timeout 1800 python -m pytest tests/test_gpu_fullsize.py -q -k "cfg3" 2>&1 | grep -i "passed\|failed\|error" | tail -3 >> $O/pytest.txt
ROUND=r04 bash profiles/tools/collect_profile.sh cfg3 > $O/collect_cfg3.log 2>&1
cd /tmp && export TMPDIR=/tmp
FW_KNOBS=1 FW_DH_CHAINS=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_one -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-schedule --no-one-chain > /root/repo/$O/cfg3_one_chain_bench_under_rocprof.json 2>/dev/null
cd /root/repo
find /tmp/prof_one -name '*kernel_stats.csv' -exec cp {} $O/cfg3_one_chain_kernel_stats.csv \;
timeout 900 python bench.py --steps 20 --warmup 1 --host-seam 2>$O/bench_cfg3.err | tail -1 > $O/bench_cfg3_n1.json
timeout 1200 python -m tests.fuzz_gpu --first 500000 --cases 1500 2>&1 | tail -1 > $O/fuzz.txt
cat $O/pytest.txt $O/fuzz.txt
python - <<'PY'
import json
l=json.loads(open("gpurun_out/r4_final_f/bench_cfg3_n1.json").read()); print("cfg3 ms %.2f other %.2f edges %d frac %.4f value %.4g launches %s"%(l["ms_per_step"], l["other_schedule"]["ms_per_step"], l["edges"], l["roofline"]["frac"], l["value"], l["kernel_launches_per_step"]))
PY
