#!/bin/bash
# r05 final collection E (final build, counter summaries of B installed): the bench lines, cfg5 (one pass), the N-rank replays of cfg3 / cfg4, the fuzz sweeps
O=gpurun_out/r5_final_e; mkdir -p $O
timeout 900 python bench.py --steps 20 --warmup 1 --host-seam 2>$O/err_cfg3.txt | tail -1 > $O/bench_cfg3.json
for cfg in cfg2 cfg4 cfg3he; do
  timeout 600 python bench.py --config $cfg --steps 10 --warmup 2 2>$O/err_$cfg.txt | tail -1 > $O/bench_$cfg.json
done
timeout 900 python bench.py --config cfg5 --steps 1 --warmup 0 --no-cpu-baseline --no-other-schedule --no-one-chain 2>$O/err_cfg5.txt | tail -1 > $O/bench_cfg5.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5_final_e/bench_*.json")):
    try:
        l=json.loads(open(f).read()); r=l["roofline"]
        print(f, "ms %.2f other %s edges %d value %.4g"%(l["ms_per_step"], (l.get("other_schedule") or {}).get("ms_per_step"), l["edges"], l["value"]), "frac %.3f bound %s valu_frac %s l0 %s"%(r["frac"], r["bound"], r.get("valu_frac"), (r.get("level0") or {}).get("frac")))
    except Exception as e: print(f, "ERR", e)
PY
bash profiles/tools/simulate_world.sh cfg3 > $O/simulate_world_cfg3.txt 2>&1; tail -6 $O/simulate_world_cfg3.txt
bash profiles/tools/simulate_world.sh cfg4 > $O/simulate_world_cfg4.txt 2>&1; tail -6 $O/simulate_world_cfg4.txt
export FW_KNOBS=1
F=gpurun_out/r5_fuzz; mkdir -p $F
timeout 900 python -m tests.fuzz_gpu --first 540000 --cases 2000 > $F/networks.txt 2>&1; tail -2 $F/networks.txt
FW_L0_MFMA=2 timeout 700 python -m tests.fuzz_gpu --first 550000 --cases 1200 > $F/networks_level0_matrix_cores_forced.txt 2>&1; tail -2 $F/networks_level0_matrix_cores_forced.txt
timeout 600 python -m tests.fuzz_gpu --subsets --first 560000 --cases 1500 > $F/subsets.txt 2>&1; tail -2 $F/subsets.txt
FW_DEV_MIN_TARGETS=8 FW_SEG_TARGET=64 timeout 600 python -m tests.fuzz_gpu --first 570000 --cases 1000 > $F/networks_small_segments.txt 2>&1; tail -2 $F/networks_small_segments.txt
