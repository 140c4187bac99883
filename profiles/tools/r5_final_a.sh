#!/bin/bash
# r05 final collection A: pytest -m gpu on the final state, bench lines of cfg3 (the driver's command shape, with the CPU leg), cfg2, cfg4, cfg3he
O=gpurun_out/r5_final_a; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -12 > $O/pytest_gpu.txt
cat $O/pytest_gpu.txt
timeout 900 python bench.py --steps 20 --warmup 1 --host-seam 2>$O/err_cfg3.txt | tail -1 > $O/bench_cfg3.json
for cfg in cfg2 cfg4 cfg3he; do
  timeout 600 python bench.py --config $cfg --steps 10 --warmup 2 2>$O/err_$cfg.txt | tail -1 > $O/bench_$cfg.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5_final_a/bench_*.json")):
    try:
        l=json.loads(open(f).read()); r=l["roofline"]
        print(f, "ms %.2f other %s edges %d"%(l["ms_per_step"], (l.get("other_schedule") or {}).get("ms_per_step"), l["edges"]), l["tests_per_step"], "frac %.3f bound %s valu_frac %s l0 %s"%(r["frac"], r["bound"], r.get("valu_frac"), (r.get("level0") or {}).get("frac")), "cpu", (l.get("cpu_baseline") or {}).get("value"))
    except Exception as e: print(f, "ERR", e)
PY
