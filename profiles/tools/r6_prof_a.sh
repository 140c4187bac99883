#!/bin/bash
# r06: counter passes of the discrete configurations on the build with statically indexed positions / inlined empty poll, cfg3he after the record fix, cfg2 host trace
export FW_KNOBS=1
mkdir -p gpurun_out/r6g
ROUND=r06 bash profiles/tools/collect_profile.sh cfg4 > gpurun_out/r6g/collect_cfg4.log 2>&1
ROUND=r06 bash profiles/tools/collect_profile.sh cfg2 > gpurun_out/r6g/collect_cfg2.log 2>&1
for lib in libfw_miprev.so libflashweave_amd.so; do FW_LIB_PATH=$PWD/flashweave.jl_amd/$lib python bench.py --config cfg3he --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('cfg3he $lib', round(l['ms_per_step'],2), l['edges'], l['tests_per_step'])" | tee -a gpurun_out/r6g/cfg3he.txt; done
FW_TRACE_HOST=1 python bench.py --config cfg2 --steps 2 --warmup 1 --no-cpu-baseline --no-other-schedule > /dev/null 2> gpurun_out/r6g/cfg2_trace.txt; tail -40 gpurun_out/r6g/cfg2_trace.txt | cut -c1-400
