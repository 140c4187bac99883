#!/bin/bash
# r05: the screened last formula of the size-5 tests (level-3 position tables, FW_L3_SCREEN) -- parity, then cfg5 at full size, A/B against
# a build without it (flashweave.jl_amd/libfw_noscr.so: make EXTRA=-DFW_L3_SCREEN=0 for fw_fz.o only)
O=gpurun_out/r5_cfg5; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_fz.py -q -x -k "size_4_5 or max_k5 or long_accepted or subsets or single_tests or oracle" 2>&1 | tail -4 > $O/pytest.txt
timeout 1500 python -m tests.fuzz_gpu --subsets --first 910000 --cases 600 2>&1 | tail -2 >> $O/pytest.txt
timeout 2400 python -m pytest tests/test_gpu_fullsize.py -q -x -k "cfg5" 2>&1 | tail -4 >> $O/pytest.txt
cat $O/pytest.txt
timeout 900 python bench.py --config cfg5 --steps 1 --warmup 0 --no-cpu-baseline --no-other-schedule --no-one-chain 2>$O/err_scr.txt | tail -1 > $O/bench_cfg5_screen.json
FW_KNOBS=1 FW_LIB_PATH=$PWD/flashweave.jl_amd/libfw_noscr.so timeout 900 python bench.py --config cfg5 --steps 1 --warmup 0 --no-cpu-baseline --no-other-schedule --no-one-chain 2>$O/err_noscr.txt | tail -1 > $O/bench_cfg5_noscreen.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5_cfg5/bench_*.json")):
    try:
        l=json.loads(open(f).read()); print(f, "s %.2f edges %d"%(l["ms_per_step"]/1e3, l["edges"]), l["tests_per_step"], {k:round(v,3) for k,v in l.get("stage_seconds_rank0").items() if k in ("conditional","level0","subsets_kernels_device")})
    except Exception as e: print(f, "ERR", e)
PY
