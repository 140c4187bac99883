"""Profiling tool: time of the discrete level-0 stage at cfg4 with parts of mi_level0_kernel disabled (FW_L0_DBG bits:
1 = no screening epilogue, 2 = no global loads, 4 = no popcount loop).  Results are invalid in those modes."""
import os, sys, time, subprocess, json
import os as _os; _os.environ.setdefault("FW_KNOBS", "1")  # the library reads FW_* knobs only when this is set
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import bench
    import flashweave_jl_amd as fw
    args = bench.parse_args([])
    cfg, _, data, _ = bench.make_input("cfg4", args)
    n, p = data.shape
    eng = fw.Engine(cfg["test_name"], n, p, max_k=3)
    eng.set_data(data)
    try:
        eng.level0()
    except Exception as e:
        pass
    t0 = time.perf_counter()
    for _ in range(3):
        try:
            eng.level0()
        except Exception:
            pass
    print(os.environ.get("FW_L0_DBG", "0"), (time.perf_counter() - t0) / 3)
else:
    for d in (os.environ.get("L0_ABLATE_SET", "0 1 2 4 3 5 7").split()):
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, FW_L0_DBG=d))
