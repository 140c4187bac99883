#!/bin/bash
# r04: matrix-core form of the discrete level-0 kernel (mi_level0_mfma_kernel): parity (level-0 tests, sharded level 0, full-size cfg4
# rows against the oracle), cfg4 with and without it, kernel durations
O=gpurun_out/r4_l; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_mi.py tests/test_gpu_dist.py -q -x 2>&1 | tail -8 > $O/pytest.txt
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -x -k "cfg4" 2>&1 | tail -8 >> $O/pytest.txt
for m in 0 1; do
  FW_KNOBS=1 FW_L0_MFMA=$m timeout 600 python bench.py --config cfg4 --steps 5 --warmup 1 --no-cpu-baseline 2>$O/err_$m.txt | tail -1 > $O/bench_cfg4_mfma$m.json
done
cd /tmp && export TMPDIR=/tmp
FW_KNOBS=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_l -o l -- python /root/repo/bench.py --config cfg4 --steps 2 --warmup 1 --no-cpu-baseline --no-other-schedule > /dev/null 2>&1
cd /root/repo
f=$(find /tmp/prof_l -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" | cut -c1-100,300-420 > $O/kernel_stats_head.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4_l/bench_*.json")):
    try:
        l=json.loads(open(f).read()); print(f, "ms %.2f other %.2f edges %d"%(l["ms_per_step"], l["other_schedule"]["ms_per_step"], l["edges"]), l.get("stage_seconds_rank0"))
    except Exception as e: print(f, "ERR", e)
PY
cat $O/pytest.txt; cat $O/kernel_stats_head.txt
