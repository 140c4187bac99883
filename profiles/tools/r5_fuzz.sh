#!/bin/bash
export FW_KNOBS=1
# r05, final kernels (level-0 matrix-core kernel rewritten, exact kernel, fz_nz on the device rounds, the discrete
# kinds' device schedule): randomised parity sweep tests/fuzz_gpu.py against the oracle, seeds disjoint from the earlier rounds'
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_fuzz; mkdir -p $O
timeout 1500 python -m tests.fuzz_gpu --first 500000 --cases 3000 > $O/networks.txt 2>&1; tail -2 $O/networks.txt
FW_L0_MFMA=2 timeout 1200 python -m tests.fuzz_gpu --first 510000 --cases 2500 > $O/networks_level0_matrix_cores_forced.txt 2>&1; tail -2 $O/networks_level0_matrix_cores_forced.txt
timeout 900 python -m tests.fuzz_gpu --subsets --first 520000 --cases 2000 > $O/subsets.txt 2>&1; tail -2 $O/subsets.txt
FW_DEV_MIN_TARGETS=8 FW_SEG_TARGET=64 timeout 900 python -m tests.fuzz_gpu --first 530000 --cases 1500 > $O/networks_small_segments.txt 2>&1; tail -2 $O/networks_small_segments.txt
