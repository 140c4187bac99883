#!/bin/bash
export FW_KNOBS=1  # the library reads FW_* knobs only when this is set
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/w8
FW_DH_CHAINS=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/w8/trace -- python $R/bench.py --config cfg3 --steps 1 --warmup 1 --no-other-schedule --no-cpu-baseline --simulate-world 8 --simulate-rank 2 --feed-forward ${FF:-1} > $R/gpurun_out/w8/bench.json 2> $R/gpurun_out/w8/err.log
cd $R
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/w8/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    for k in ("dh_step", "dh_plan", "dh_fill", "dh_compact", "fz_subsets_seg_kernel<false, false, true>", "fz_subsets_seg_kernel<false, false, false>"):
        if k in n: return k.replace("fz_subsets_seg_kernel","seg")
    return None
seq = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
seq = [s for s in seq if s[0]]
# last pass = the timed step of the simulated rank: take the last 40 % of the kernels
seq = seq[int(len(seq)*0.75):]
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
for a, b in zip(seq, seq[1:]):
    dur[a[0]].append(a[2]-a[1]); gap[a[0] + " -> " + b[0]].append(b[1]-a[2])
for k, v in dur.items(): print("dur", k, len(v), round(sum(v)/len(v)/1e3, 2), "us  total ms", round(sum(v)/1e6,2))
for k, v in sorted(gap.items(), key=lambda kv: -len(kv[1]))[:8]: print("gap", k, len(v), round(sum(v)/len(v)/1e3, 2), "us total ms", round(sum(v)/1e6,2))
tot = seq[-1][2] - seq[0][1]
print("span ms", tot/1e6, "kernels", len(seq))
# histogram of seg durations
import statistics
sd = sorted(dur.get("seg<false, false, true>", []))
if sd:
    n=len(sd); print("seg pct us", [round(sd[int(n*q)]/1e3,1) for q in (0.1,0.25,0.5,0.75,0.9,0.99)])
PY
rm -rf gpurun_out/w8/trace
