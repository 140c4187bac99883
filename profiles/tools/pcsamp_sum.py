"""Aggregate rocprofv3 PC-sampling CSVs (host_trap or stochastic) into a histogram small enough to track:
python pcsamp_sum.py <dir> [kernel-substring] -> JSON on stdout.  Samples are grouped by (instruction text, source comment) and, for
the stochastic method, by issue / stall columns; the kernel-trace CSV next to it (if any) maps dispatch ids to kernel names."""
import csv, glob, json, sys, collections
csv.field_size_limit(1 << 30)
d = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ""
disp = {}
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        disp[r.get("Dispatch_Id")] = r.get("Kernel_Name", "")
files = glob.glob(d + "/**/*pc_sampling*.csv", recursive=True)
out = {"files": files, "columns": None, "samples": 0, "by_kernel": {}, "by_inst": [], "by_stall": {}, "by_type": {}}
cnt = collections.Counter()
stall = collections.Counter()
itype = collections.Counter()
kern = collections.Counter()
issued = collections.Counter()
for f in files:
    rd = csv.DictReader(open(f))
    out["columns"] = rd.fieldnames
    for r in rd:
        k = disp.get(r.get("Dispatch_Id"), "?")
        kern[k.split("(")[0][-80:]] += 1
        if want and want not in k:
            continue
        out["samples"] += 1
        cnt[(r.get("Instruction", ""), r.get("Instruction_Comment", ""))] += 1
        if "Stall_Reason" in r:
            stall[r.get("Stall_Reason", "")] += 1
        if "Instruction_Type" in r:
            itype[r.get("Instruction_Type", "")] += 1
        if "Wave_Issued_Instruction" in r:
            issued[r.get("Wave_Issued_Instruction", "")] += 1
out["by_kernel"] = dict(kern.most_common(40))
out["by_stall"] = dict(stall)
out["by_type"] = dict(itype)
out["issued"] = dict(issued)
out["by_inst"] = [{"inst": i, "src": c, "n": n} for (i, c), n in cnt.most_common(6000)]
print(json.dumps(out))
