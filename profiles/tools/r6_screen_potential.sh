#!/bin/bash
# r06: how many fast-loop iterations of cfg3 could a cheap conservative screen decide?  (-DFW_FZ_FASTDBG=4 build: libfw_dbg4.so)
export FW_KNOBS=1
mkdir -p gpurun_out/r6h
FW_LIB_PATH=$PWD/flashweave.jl_amd/libfw_dbg4.so timeout 600 python bench.py --config cfg3 --steps 1 --warmup 0 --no-cpu-baseline --no-other-schedule --no-one-chain > gpurun_out/r6h/bench.json 2> gpurun_out/r6h/err.txt
grep "cheap-screen\|fast loop\|segments (table" gpurun_out/r6h/err.txt | tail -6
timeout 600 python -m pytest tests/test_gpu_determinism.py -m gpu -q -x 2>&1 | tail -3
