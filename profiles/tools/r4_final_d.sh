#!/bin/bash
# r04 final collection D: cfg4 after the matrix-core level-0 kernel -- kernel trace + PMC passes (collect_profile.sh), the kernel's own
# phase counters (FW_L0_VERBOSE), bench lines of cfg4 / cfg2 with the CPU leg, pytest -m gpu on the final build
O=gpurun_out/r4_final_d; mkdir -p $O
ROUND=r04 bash profiles/tools/collect_profile.sh cfg4 > $O/collect_cfg4.log 2>&1
FW_KNOBS=1 FW_L0_VERBOSE=1 timeout 600 python profiles/tools/l0_ablate.py child 2>&1 | grep "fw\]" | tail -2 > $O/level0_phase_cycles.txt
for m in 0 1; do FW_L0_MFMA=$m L0_ABLATE_SET="0" timeout 600 python profiles/tools/l0_ablate.py 2>&1 | grep "^0" | sed "s/^0/FW_L0_MFMA=$m level-0 seconds/" >> $O/level0_phase_cycles.txt; done
timeout 900 python bench.py --config cfg4 --steps 10 --warmup 1 2>$O/bench_cfg4.err | tail -1 > $O/bench_cfg4_n1.json
timeout 900 python bench.py --config cfg2 --steps 20 --warmup 2 2>$O/bench_cfg2.err | tail -1 > $O/bench_cfg2_n1.json
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -5 > $O/pytest_gpu.txt
cat $O/level0_phase_cycles.txt $O/pytest_gpu.txt
python - <<'PY'
import json
for c in ("cfg4","cfg2"):
    try:
        l=json.loads(open("gpurun_out/r4_final_d/bench_%s_n1.json"%c).read()); print(c,"ms %.2f other %.2f edges %d"%(l["ms_per_step"], l["other_schedule"]["ms_per_step"], l["edges"]), l["roofline"]["bound"], l["roofline"]["frac"])
    except Exception as e: print(c,"ERR",e)
PY
