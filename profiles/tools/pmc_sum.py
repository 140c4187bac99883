"""Sum rocprofv3 --pmc counter_collection CSVs per kernel name: python pmc_sum.py <dir> -> JSON on stdout."""
import csv, glob, json, sys, collections
out = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        out[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
print(json.dumps({k: dict(v, dispatches=len(disp[k])) for k, v in out.items()}, indent=1))
