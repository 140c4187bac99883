"""Turn the raw output of collect_profile.sh (gpurun_out/prof_<round>_<cfg>/) into the tracked files profiles/<round>_<cfg>_*:
python profiles/tools/install_profile.py <cfg> <kernel-name-substring> [round, default r03]   (run from the repo root)"""
import csv, json, shutil, sys

cfg, kname = sys.argv[1], sys.argv[2]
RND = sys.argv[3] if len(sys.argv) > 3 else "r05"
P = "gpurun_out/prof_%s_%s/" % (RND, cfg)
sq, f, w = (json.load(open(P + n)) for n in ("pmc_sq.json", "pmc_f.json", "pmc_w.json"))
import os
mix = [json.load(open(P + n)) for n in ("pmc_m1.json", "pmc_m2.json") if os.path.exists(P + n)]
bench = json.loads([l for l in open(P + "bench_under_rocprof.json") if l.startswith("{")][-1])
allk = {}
for d in [sq, f, w] + mix:
    for k, v in d.items():
        allk.setdefault(k, {}).update(v)
keys = [k for k in allk if kname in k]
if not keys:  # pmc_sum.py cuts kernel names at the first "(": "void (anonymous namespace)::dh_mi_target_kernel<..>" becomes "void "
    keys = [k for k in allk if k.strip() in ("", "void")]
K = {}
for k in keys:  # several instantiations of one kernel template: add them up
    for c, v in allk[k].items():
        K[c] = K.get(c, 0.0) + v
ev = bench["tests_per_step"]["conditional_evaluated"]
alg = bench["roofline"]["alg_bytes_per_launch"] * bench["roofline"]["launches"] / bench["steps"]
fetch, wr = K["FETCH_SIZE"] * 1024.0, K["WRITE_SIZE"] * 1024.0
nonempty = bench["roofline"]["launches"] / bench["steps"]
summ = {
    "kernels": keys, "dispatches": int(K["dispatches"]), "nonempty_launches": nonempty, "evaluated_tests": ev,
    "FETCH_SIZE_KB": K["FETCH_SIZE"], "WRITE_SIZE_KB": K["WRITE_SIZE"], "fetch_bytes_raw": fetch, "write_bytes_raw": wr,
    "fetch_bytes_note": "FETCH_SIZE counts 64-B fabric requests; the guide's x2 correction is calibrated for wide (16 B/lane) "
                        "streaming reads only - these kernels gather 4-byte words / 256-byte rows, so the raw value is reported",
    "fetch_bytes_per_launch": (fetch + wr) / nonempty,
    "l2_hit_rate": K["TCC_HIT_sum"] / K["TCC_REQ_sum"],
    "valu_wave_insts_per_test": K["SQ_INSTS_VALU"] / ev, "salu_wave_insts_per_test": K["SQ_INSTS_SALU"] / ev,
    "lds_wave_insts_per_test": K["SQ_INSTS_LDS"] / ev, "vmem_rd_wave_insts_per_test": K["SQ_INSTS_VMEM_RD"] / ev,
    "wait_inst_any_frac_of_wave_cycles": K["SQ_WAIT_INST_ANY"] / K["SQ_WAVE_CYCLES"],
    "active_inst_valu_frac_of_wave_cycles": K["SQ_ACTIVE_INST_VALU"] / K["SQ_WAVE_CYCLES"],
    "algorithmic_bytes": alg, "fabric_over_algorithmic": (fetch + wr) / alg,
    # resident wavefronts per SIMD the kernel is COMPILED for (the -Rpass-analysis=kernel-resource-usage remark of the instantiation:
    # 4 for the size-3 segment kernel, 3 for the long-list max_k 4-5 variant, 1 for the persistent discrete kernel); bench.py multiplies
    # SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES by it to get the VALU-busy share of a SIMD
    "waves_per_simd": 1 if "dh_mi" in kname else (3 if bench["config"]["workload"].find("max_k=5") >= 0 or bench["config"]["workload"].find("max_k=4") >= 0 else 4),
}
MIXC = {"f64_add": "SQ_INSTS_VALU_ADD_F64", "f64_mul": "SQ_INSTS_VALU_MUL_F64", "f64_fma": "SQ_INSTS_VALU_FMA_F64", "f64_trans": "SQ_INSTS_VALU_TRANS_F64",
        "f32_add": "SQ_INSTS_VALU_ADD_F32", "f32_mul": "SQ_INSTS_VALU_MUL_F32", "f32_fma": "SQ_INSTS_VALU_FMA_F32", "f32_trans": "SQ_INSTS_VALU_TRANS_F32",
        "int32": "SQ_INSTS_VALU_INT32", "int64": "SQ_INSTS_VALU_INT64", "cvt": "SQ_INSTS_VALU_CVT"}
if all(c in K for c in MIXC.values()):
    m = {k: K[c] / ev for k, c in MIXC.items()}
    m["other"] = max(0.0, summ["valu_wave_insts_per_test"] - sum(m.values()))
    summ["valu_mix_wave_insts_per_test"] = m
out = {
    "command": "profiles/tools/collect_profile.sh %s: timeout 900 rocprofv3 --kernel-trace --pmc <counters> --output-format csv -- python "
               "bench.py --config %s --steps 1 --warmup 0 --no-cpu-baseline --no-other-schedule --no-one-chain (three separate passes: SQ_*; "
               "FETCH_SIZE+TCC_HIT_sum; TCC_MISS_sum+TCC_REQ_sum+WRITE_SIZE), summed per kernel with profiles/tools/pmc_sum.py" % (cfg, cfg),
    "workload": bench["config"]["workload"], "schedule": {k: bench["config"][k] for k in ("feed_forward", "round_size")},
    "counters_per_kernel_sum_over_dispatches": allk,
    ("fz_subsets_seg_kernel" if "fz" in kname else ("dh_mi_target_kernel" if "dh_mi" in kname else "mi_subsets_seg_kernel")): summ,
}
json.dump(out, open("profiles/%s_%s_pmc_summary.json" % (RND, cfg), "w"), indent=1)
shutil.copy(P + "kernel_stats.csv", "profiles/%s_%s_kernel_stats.csv" % (RND, cfg))
json.dump(bench, open("profiles/%s_%s_bench_under_rocprof.json" % (RND, cfg), "w"))
print(json.dumps({k: v for k, v in summ.items() if k != "fetch_bytes_note"}, indent=1))
print("under rocprof: ms %.1f launch_us %.1f launches %d" % (bench["ms_per_step"], bench["roofline"]["avg_launch_us"], bench["roofline"]["launches"]))
for r in csv.DictReader(open(P + "kernel_stats.csv")):
    if float(r["TotalDurationNs"]) > 2e6:
        print(r["Name"][:70], r["Calls"], "%.3f ms total" % (float(r["TotalDurationNs"]) / 1e6), "avg %.1f us" % (float(r["AverageNs"]) / 1e3), r["Percentage"])
