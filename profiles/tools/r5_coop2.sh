#!/bin/bash
# r05: kernel-level view of the cooperative round: one chain, rocprofv3 kernel trace, FW_DH_FUSE = 1 / 0; then the two-chain headline
O=gpurun_out/r5_coop2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for f in 1 0; do
  FW_KNOBS=1 FW_DH_CHAINS=1 FW_DH_FUSE=$f rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_fuse$f -o t -- python $GRAFT_REPO_ROOT/bench.py --config cfg3 --steps 2 --warmup 1 --no-cpu-baseline --no-other-schedule --no-one-chain > $GRAFT_REPO_ROOT/$O/one_chain_fuse$f.json 2>$GRAFT_REPO_ROOT/$O/err1_$f.txt
  python - <<PY
import csv,glob
for f in glob.glob("$GRAFT_REPO_ROOT/$O/prof_fuse$f/**/*kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:8]: print("fuse$f", r["Name"][:60], r["Calls"], r["TotalDurationNs"], r["AverageNs"])
PY
done
cd $GRAFT_REPO_ROOT
for f in 1 0; do
  FW_KNOBS=1 FW_DH_FUSE=$f timeout 600 python bench.py --config cfg3 --steps 8 --warmup 1 --no-cpu-baseline 2>$O/err_$f.txt | tail -1 > $O/bench_cfg3_fuse$f.json
done
for g in 32 64 256; do
  FW_KNOBS=1 FW_DH_COOP_GRID=$g timeout 600 python bench.py --config cfg3 --steps 8 --warmup 1 --no-cpu-baseline --no-other-schedule 2>/dev/null | tail -1 > $O/bench_cfg3_grid$g.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5_coop2/*.json")):
    try:
        l=json.loads(open(f).read().strip().split("\n")[-1]); print(f, "ms %.2f edges %d"%(l["ms_per_step"], l["edges"]), l["roofline"].get("step_seconds_of_that_pass"))
    except Exception as e: print(f, "ERR", e)
PY
rm -rf $O/prof_fuse*/*/*.db 2>/dev/null
