"""Idle time between consecutive kernels of one stream, from a rocprofv3 --kernel-trace CSV:
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline
    python profiles/tools/gap_analysis.py $(find /tmp/gp -name '*kernel_trace.csv')"""
import csv, sys, collections

import re


def short(name):  # "void ns::(anonymous namespace)::dh_step_kernel(args...)" -> "dh_step_kernel"
    name = name.replace("(anonymous namespace)::", "")
    m = re.search(r"([A-Za-z_][A-Za-z_0-9]*)(<[^(]*>)?\(", name)
    return (m.group(1) if m else name)[:28]


rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# keep the last pass: everything after the last correlation GEMM
last = max(i for i, r in enumerate(rows) if "fz_cor_gemm" in r["Kernel_Name"]) if any("fz_cor_gemm" in r["Kernel_Name"] for r in rows) else 0
rows = rows[last:]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
wall = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
gaps = collections.defaultdict(list)
for a, b in zip(rows[:-1], rows[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    key = "%s -> %s" % (short(a["Kernel_Name"]), short(b["Kernel_Name"]))
    gaps[key].append(g)
# union of the kernel intervals (concurrent streams overlap: the sum of the durations can exceed the wall time)
ivs = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
union, cur_s, cur_e = 0, ivs[0][0], ivs[0][1]
for s0, e0 in ivs[1:]:
    if s0 > cur_e:
        union += cur_e - cur_s
        cur_s, cur_e = s0, e0
    else:
        cur_e = max(cur_e, e0)
union += cur_e - cur_s
seg = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "subsets_seg_kernel" in r["Kernel_Name"])
seg_union = 0
if seg:
    cs, ce = seg[0]
    for s0, e0 in seg[1:]:
        if s0 > ce:
            seg_union += ce - cs
            cs, ce = s0, e0
        else:
            ce = max(ce, e0)
    seg_union += ce - cs
print("kernels %d  wall %.1f ms  sum of durations %.1f ms  GPU busy (union) %.1f ms  idle %.1f ms  segment kernel busy (union) %.1f ms over %d launches"
      % (len(rows), wall / 1e6, busy / 1e6, union / 1e6, (wall - union) / 1e6, seg_union / 1e6, len(seg)))
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:12]:
    v2 = sorted(v)
    print("%-62s n %5d  sum %7.2f ms  median %6.2f us  p90 %6.2f us" % (k, len(v), sum(v) / 1e6, v2[len(v2) // 2] / 1e3, v2[int(len(v2) * 0.9)] / 1e3))
