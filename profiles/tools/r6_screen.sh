#!/bin/bash
# r06: the cheap screen of the size-3 fast loop: (1) validation build (-DFW_FZ_FASTDBG=5: the screen decides nothing, every lane it would skip is checked against the exact value),
# (2) A/B against the build without it (-DFW_FZ_SCREEN=0) on one box, (3) the parity suites
export FW_KNOBS=1
O=gpurun_out/r6_screen; mkdir -p $O
FW_LIB_PATH=$PWD/flashweave.jl_amd/libfw_dbg5.so timeout 900 python bench.py --config cfg3 --steps 1 --warmup 0 --no-cpu-baseline --no-one-chain > $O/val.json 2> $O/val_err.txt
grep "cheap screen" $O/val_err.txt | tail -2 | tee $O/validation.txt
FW_LIB_PATH=$PWD/flashweave.jl_amd/libfw_dbg5.so timeout 900 python bench.py --config cfg3he --steps 1 --warmup 0 --no-cpu-baseline --no-one-chain > $O/val_he.json 2> $O/val_he_err.txt
grep "cheap screen" $O/val_he_err.txt | tail -1 | sed 's/^/cfg3he /' | tee -a $O/validation.txt
bash profiles/tools/r6_ab.sh cfg3 2 libfw_scr0.so libflashweave_amd.so
timeout 1200 python -m pytest tests/test_gpu_fz.py tests/test_gpu_fznz.py tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k "cfg3" 2>&1 | tail -3
