#!/bin/bash
# r04 final collection E (final build: MX-fp4 level 0 with double-buffered stages, fz_nz without a matrix, byte row form): pytest -m gpu,
# cfg4 kernel trace + PMC passes, the level-0 kernel's phase counters, bench lines of cfg3 (headline, with the CPU leg) / cfg4 / cfg2 / cfg3he
O=gpurun_out/r4_final_e; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | grep -i "passed\|failed\|error" | tail -5 > $O/pytest_gpu.txt
ROUND=r04 bash profiles/tools/collect_profile.sh cfg4 > $O/collect_cfg4.log 2>&1
FW_KNOBS=1 FW_L0_VERBOSE=1 timeout 600 python profiles/tools/l0_ablate.py child 2>&1 | grep "fw\]" | tail -2 > $O/level0_phase_cycles.txt
for m in 0 1; do FW_L0_MFMA=$m L0_ABLATE_SET="0" timeout 600 python profiles/tools/l0_ablate.py 2>&1 | grep "^0" | sed "s/^0/FW_L0_MFMA=$m level-0 seconds/" >> $O/level0_phase_cycles.txt; done
timeout 900 python bench.py --config cfg4 --steps 10 --warmup 1 2>$O/bench_cfg4.err | tail -1 > $O/bench_cfg4_n1.json
timeout 900 python bench.py --config cfg2 --steps 20 --warmup 2 2>$O/bench_cfg2.err | tail -1 > $O/bench_cfg2_n1.json
timeout 900 python bench.py --config cfg3he --steps 10 --warmup 1 2>$O/bench_cfg3he.err | tail -1 > $O/bench_cfg3he_n1.json
timeout 900 python bench.py --steps 20 --warmup 1 --host-seam 2>$O/bench_cfg3.err | tail -1 > $O/bench_cfg3_n1.json
cat $O/level0_phase_cycles.txt $O/pytest_gpu.txt
python - <<'PY'
import json
for c in ("cfg3","cfg4","cfg2","cfg3he"):
    try:
        l=json.loads(open("gpurun_out/r4_final_e/bench_%s_n1.json"%c).read()); print(c,"ms %.2f other %.2f edges %d"%(l["ms_per_step"], l["other_schedule"]["ms_per_step"], l["edges"]), l["roofline"]["bound"], round(l["roofline"]["frac"],4), "value %.3g"%l["value"])
    except Exception as e: print(c,"ERR",e)
PY
