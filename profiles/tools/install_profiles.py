"""Turn the raw output of collect_cfg3.sh (gpurun_out/prof_cfg3/) and the bench lines under gpurun_out/final/ into the
tracked files profiles/r01_*: python profiles/tools/install_profiles.py   (run from the repo root)"""
import csv, json, os, shutil

P = "gpurun_out/prof_cfg3/"
sq, f, w = (json.load(open(P + n)) for n in ("pmc_sq.json", "pmc_f.json", "pmc_w.json"))
bench = json.load(open(P + "bench_under_rocprof.json"))
allk = {}
for d in (sq, f, w):
    for k, v in d.items():
        allk.setdefault(k, {}).update(v)
key = [k for k in allk if "fz_subsets_seg_kernel<false, false, true>" in k][0]
K = allk[key]
ev = bench["tests_per_step"]["conditional_evaluated"]
alg = bench["roofline"]["alg_bytes_per_launch"] * bench["roofline"]["launches"] / bench["steps"]
fetch, wr = K["FETCH_SIZE"] * 1024.0, K["WRITE_SIZE"] * 1024.0
nonempty = bench["roofline"]["launches"] / bench["steps"]
summ = {
    "kernel": key, "dispatches": int(K["dispatches"]), "nonempty_launches": nonempty, "evaluated_tests": ev,
    "FETCH_SIZE_KB": K["FETCH_SIZE"], "WRITE_SIZE_KB": K["WRITE_SIZE"], "fetch_bytes_raw": fetch, "write_bytes_raw": wr,
    "fetch_bytes_note": "FETCH_SIZE counts 64-B fabric requests; the guide's x2 correction is calibrated for wide (16 B/lane) "
                        "streaming reads only - these are 4-byte gathers plus one 4-byte read per LDS-table entry, so the raw "
                        "value is reported (x2 = upper bound if every miss is a 128-B line)",
    "fetch_bytes_per_launch": (fetch + wr) / nonempty,
    "l2_hit_rate": K["TCC_HIT_sum"] / K["TCC_REQ_sum"],
    "valu_wave_insts_per_test": K["SQ_INSTS_VALU"] / ev, "salu_wave_insts_per_test": K["SQ_INSTS_SALU"] / ev,
    "lds_wave_insts_per_test": K["SQ_INSTS_LDS"] / ev, "vmem_rd_wave_insts_per_test": K["SQ_INSTS_VMEM_RD"] / ev,
    "wait_inst_any_frac_of_wave_cycles": K["SQ_WAIT_INST_ANY"] / K["SQ_WAVE_CYCLES"],
    "active_inst_valu_frac_of_wave_cycles": K["SQ_ACTIVE_INST_VALU"] / K["SQ_WAVE_CYCLES"],
    "algorithmic_bytes": alg, "fabric_over_algorithmic": (fetch + wr) / alg,
}
out = {
    "command": "profiles/tools/collect_cfg3.sh: timeout 600 rocprofv3 --kernel-trace --pmc <counters> --output-format csv -- python "
               "bench.py --config cfg3 --steps 1 --warmup 0 --no-cpu-baseline (three separate passes: SQ_*; FETCH_SIZE+TCC_HIT_sum; "
               "TCC_MISS_sum+TCC_REQ_sum+WRITE_SIZE), summed per kernel with profiles/tools/pmc_sum.py",
    "workload": bench["config"]["workload"], "counters_per_kernel_sum_over_dispatches": allk, "fz_subsets_seg_kernel": summ,
}
json.dump(out, open("profiles/r01_cfg3_fz_pmc_summary.json", "w"), indent=1)
shutil.copy(P + "kernel_stats.csv", "profiles/r01_cfg3_fz_kernel_stats.csv")
shutil.copy(P + "bench_under_rocprof.json", "profiles/r01_cfg3_fz_bench_under_rocprof.json")
for a, b in (("bench_cfg3", "r01_bench_cfg3_n1"), ("bench_cfg2", "r01_bench_cfg2_n1"), ("bench_cfg4", "r01_bench_cfg4_n1"),
             ("cfg4_bench_under_rocprof", "r01_cfg4_bench_under_rocprof")):
    if os.path.exists("gpurun_out/final/%s.json" % a) and os.path.getsize("gpurun_out/final/%s.json" % a) > 0:
        shutil.copy("gpurun_out/final/%s.json" % a, "profiles/%s.json" % b)
if os.path.exists("gpurun_out/final/cfg4_kernel_stats.csv"):
    shutil.copy("gpurun_out/final/cfg4_kernel_stats.csv", "profiles/r01_cfg4_mi_nz_kernel_stats.csv")
print(json.dumps({k: v for k, v in summ.items() if k != "fetch_bytes_note"}, indent=1))
for n in ("bench_cfg3", "bench_cfg2", "bench_cfg4"):
    fn = "gpurun_out/final/%s.json" % n
    if os.path.exists(fn) and os.path.getsize(fn) > 0:
        b = json.load(open(fn))
        print(n, "value %.4g" % b["value"], "ms %.1f" % b["ms_per_step"], "edges", b["edges"], "frac %.3f" % b["roofline"]["frac"],
              "launch_us %.1f" % b["roofline"]["avg_launch_us"], "launches", b["roofline"]["launches"],
              "cpu", (b["cpu_baseline"] or {}).get("value"), b["tests_per_step"])
print("under rocprof: ms %.1f launch_us %.1f launches %d" % (bench["ms_per_step"], bench["roofline"]["avg_launch_us"], bench["roofline"]["launches"]))
for fn in (P + "kernel_stats.csv", "gpurun_out/final/cfg4_kernel_stats.csv"):
    if not os.path.exists(fn):
        continue
    for r in csv.DictReader(open(fn)):
        if float(r["TotalDurationNs"]) > 2e6:
            print(r["Name"][:60], r["Calls"], "%.3f ms total" % (float(r["TotalDurationNs"]) / 1e6), "avg %.1f us" % (float(r["AverageNs"]) / 1e3), r["Percentage"])
