#!/bin/bash
export FW_KNOBS=1
# r06, after the max_k 4-5 kernels took the local matrices: randomised whole networks with fz, max_k 4 / 5 (tests/fuzz_gpu.py --highk) against the oracle, seeds 650000-...;
# once with the defaults, once with tiny segments and matrices for every target, once without matrices; then the usual sweep on the final library (seeds 660000-...)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6_fuzz2; mkdir -p $O
timeout 700 python -m tests.fuzz_gpu --highk --first 650000 --cases 400 > $O/highk.txt 2>&1; tail -2 $O/highk.txt
FW_DEV_MIN_TARGETS=1 FW_SEG_TARGET=64 FW_FZ_TMAT=1 timeout 700 python -m tests.fuzz_gpu --highk --first 651000 --cases 400 > $O/highk_small_segments.txt 2>&1; tail -2 $O/highk_small_segments.txt
FW_FZ_TMAT=0 timeout 500 python -m tests.fuzz_gpu --highk --first 652000 --cases 250 > $O/highk_no_matrices.txt 2>&1; tail -2 $O/highk_no_matrices.txt
timeout 700 python -m tests.fuzz_gpu --first 660000 --cases 1200 > $O/networks.txt 2>&1; tail -2 $O/networks.txt
