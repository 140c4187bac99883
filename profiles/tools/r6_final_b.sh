#!/bin/bash
# r06 final collection B (after the PMC summaries of A are installed under profiles/, which the bench lines read): the bench lines; the full-size module once more (timing of the reordered tests)
O=gpurun_out/r6_final_b; mkdir -p $O
timeout 900 python bench.py --steps 20 --warmup 1 2>$O/err_cfg3.txt | tail -1 > $O/bench_cfg3.json
for cfg in cfg2 cfg4 cfg3he; do timeout 600 python bench.py --config $cfg --steps 10 --warmup 2 2>$O/err_$cfg.txt | tail -1 > $O/bench_$cfg.json; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r6_final_b/bench_*.json")):
    try:
        l=json.loads(open(f).read()); r=l["roofline"]
        print(f, "ms %.2f other %s edges %d"%(l["ms_per_step"], (l.get("other_schedule") or {}).get("ms_per_step"), l["edges"]), "frac %.3f bound %s valu_frac %s traffic %s l0 %s w/r %s cpu %s"%(r["frac"], r["bound"], r.get("valu_frac"), r.get("traffic"), (r.get("level0") or {}).get("frac"), r.get("write_over_result_bytes"), (l.get("cpu_baseline") or {}).get("value")))
    except Exception as e: print(f, "ERR", e)
PY

