#!/bin/bash
# what the slowest rank of an N-rank cfg3 job costs, measured on ONE GPU: bench.py --simulate-world N --simulate-rank -1 records a
# single-rank pass, then times every rank's share in turn with the recorded all-gathers replayed (same whitelists as among N real
# ranks; the exchange itself is not included).  Prints: N, slowest rank ms (headline schedule / feed_forward 0), per-rank lists.
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/simulate_world
mkdir -p $O
cd $R
CFG=${1:-cfg3}
shift
for N in 1 2 4 8; do
  if [ $N = 1 ]; then SW=""; else SW="--simulate-world $N --simulate-rank -1"; fi
  timeout 900 python bench.py --config $CFG $SW --steps 3 --warmup 1 --no-cpu-baseline "$@" > $O/${CFG}_n$N.json 2> $O/${CFG}_n$N.err
  python - $O/${CFG}_n$N.json $N <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
o = d.get("other_schedule") or {}
f = lambda m: [round(v, 1) for v in (m.get("simulated_world") or {}).get("ms_per_step_by_rank", [])]
print("N=%s  ff=1: %.1f ms %s   ff=0: %s ms %s" % (sys.argv[2], d["ms_per_step"], f(d), round(o.get("ms_per_step", 0), 1), f(o)))
PY
done
