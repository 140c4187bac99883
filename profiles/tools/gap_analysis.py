"""Idle time between consecutive kernels of one stream, from a rocprofv3 --kernel-trace CSV:
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline
    python profiles/tools/gap_analysis.py $(find /tmp/gp -name '*kernel_trace.csv')"""
import csv, sys, collections

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
    key = "%s -> %s" % (a["Kernel_Name"].split("(")[0][-28:], b["Kernel_Name"].split("(")[0][-28:])
    gaps[key].append(g)
print("kernels %d  wall %.1f ms  busy %.1f ms  idle %.1f ms" % (len(rows), wall / 1e6, busy / 1e6, (wall - busy) / 1e6))
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:12]:
    v2 = sorted(v)
    print("%-62s n %5d  sum %7.2f ms  median %6.2f us  p90 %6.2f us" % (k, len(v), sum(v) / 1e6, v2[len(v2) // 2] / 1e3, v2[int(len(v2) * 0.9)] / 1e3))
