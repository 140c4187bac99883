#!/bin/bash
export FW_KNOBS=1
# r04, final kernels (matrix-core level 0, row form of the counting phase, generic discrete form, fz_nz without a matrix are in): randomised
# parity sweep tests/fuzz_gpu.py, seeds disjoint from the earlier sweeps
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_fuzz3; mkdir -p $O
timeout 1500 python -m tests.fuzz_gpu --first 400000 --cases 2500 > $O/networks.txt 2>&1; tail -2 $O/networks.txt
FW_L0_MFMA=2 timeout 1200 python -m tests.fuzz_gpu --first 410000 --cases 2000 > $O/networks_level0_matrix_cores_forced.txt 2>&1; tail -2 $O/networks_level0_matrix_cores_forced.txt
timeout 900 python -m tests.fuzz_gpu --subsets --first 420000 --cases 2000 > $O/subsets.txt 2>&1; tail -2 $O/subsets.txt
FW_MI_ROWK=1 FW_DEV_MIN_TARGETS=8 timeout 900 python -m tests.fuzz_gpu --first 430000 --cases 1500 > $O/networks_row_form_k1.txt 2>&1; tail -2 $O/networks_row_form_k1.txt
