#!/bin/bash
export FW_KNOBS=1
O=gpurun_out/r6_screen; mkdir -p $O
FW_LIB_PATH=$PWD/flashweave.jl_amd/libfw_dbg5.so timeout 900 python bench.py --config cfg3 --steps 1 --warmup 0 --no-cpu-baseline --no-one-chain --no-other-schedule > $O/val.json 2> $O/val_err.txt
grep "cheap screen" $O/val_err.txt | tail -1 | tee $O/validation2.txt
