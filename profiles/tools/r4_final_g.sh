#!/bin/bash
# r04 final collection G (last build: first windows 2 048 / 32 768 for fz, 48 sequential tests before a board for the discrete kinds):
# pytest -m gpu, cfg4 kernel trace + PMC passes, bench lines cfg4 / cfg2 / cfg3he with the CPU leg
O=gpurun_out/r4_final_g; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | grep -i "passed\|failed\|error" | tail -5 > $O/pytest_gpu.txt
ROUND=r04 bash profiles/tools/collect_profile.sh cfg4 > $O/collect_cfg4.log 2>&1
timeout 900 python bench.py --config cfg4 --steps 10 --warmup 1 2>$O/bench_cfg4.err | tail -1 > $O/bench_cfg4_n1.json
timeout 900 python bench.py --config cfg2 --steps 20 --warmup 2 2>$O/bench_cfg2.err | tail -1 > $O/bench_cfg2_n1.json
cat $O/pytest_gpu.txt
python - <<'PY'
import json
for c in ("cfg4","cfg2"):
    try:
        l=json.loads(open("gpurun_out/r4_final_g/bench_%s_n1.json"%c).read()); print(c,"ms %.2f other %.2f edges %d"%(l["ms_per_step"], l["other_schedule"]["ms_per_step"], l["edges"]), l["roofline"]["bound"], round(l["roofline"]["frac"],4), "value %.3g other %.3g"%(l["value"], l["other_schedule"]["value"]))
    except Exception as e: print(c,"ERR",e)
PY
