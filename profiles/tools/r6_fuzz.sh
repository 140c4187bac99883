#!/bin/bash
export FW_KNOBS=1
# r06, final kernels (cheap screen, per-target local matrices with transposed gathers, chunks of 16 384 ranks, static positions / inlined poll in the persistent discrete kernel):
# randomised parity sweeps tests/fuzz_gpu.py against the oracle, seeds 600000-640999 (disjoint from every earlier sweep)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6_fuzz; mkdir -p $O
timeout 1500 python -m tests.fuzz_gpu --first 600000 --cases 2500 > $O/networks.txt 2>&1; tail -2 $O/networks.txt
FW_FZ_TMAT=1 timeout 900 python -m tests.fuzz_gpu --first 610000 --cases 1500 > $O/networks_local_matrices_for_every_target.txt 2>&1; tail -2 $O/networks_local_matrices_for_every_target.txt
timeout 900 python -m tests.fuzz_gpu --subsets --first 620000 --cases 2000 > $O/subsets.txt 2>&1; tail -2 $O/subsets.txt
FW_DEV_MIN_TARGETS=8 FW_SEG_TARGET=64 FW_FZ_TMAT=1 timeout 900 python -m tests.fuzz_gpu --first 630000 --cases 1500 > $O/networks_small_segments.txt 2>&1; tail -2 $O/networks_small_segments.txt
timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k "independent_of_schedule" 2>&1 | tail -3
