#!/bin/bash
# r05 final collection D (after the PMC summaries of B are installed under profiles/, which the bench lines read for `traffic` / `valu_frac`):
# the bench lines again, then cfg5's kernel trace + counter passes (one pass each)
O=gpurun_out/r5_final_d; mkdir -p $O
timeout 900 python bench.py --steps 20 --warmup 1 --host-seam 2>$O/err_cfg3.txt | tail -1 > $O/bench_cfg3.json
for cfg in cfg2 cfg4 cfg3he; do
  timeout 600 python bench.py --config $cfg --steps 10 --warmup 2 2>$O/err_$cfg.txt | tail -1 > $O/bench_$cfg.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5_final_d/bench_*.json")):
    try:
        l=json.loads(open(f).read()); r=l["roofline"]
        print(f, "ms %.2f other %s edges %d"%(l["ms_per_step"], (l.get("other_schedule") or {}).get("ms_per_step"), l["edges"]), "frac %.3f bound %s valu_frac %s traffic %s l0 %s"%(r["frac"], r["bound"], r.get("valu_frac"), r.get("traffic"), (r.get("level0") or {}).get("frac")))
    except Exception as e: print(f, "ERR", e)
PY
ROUND=r05 STATS_STEPS=1 STATS_WARMUP=0 PROF_TIMEOUT=1500 bash profiles/tools/collect_profile.sh cfg5 > gpurun_out/collect_cfg5.log 2>&1
ls gpurun_out/prof_r05_cfg5
