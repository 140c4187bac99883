#!/bin/bash
export FW_KNOBS=1  # the library reads FW_* knobs only when this is set
# long randomised parity sweep on the final kernels of round 3 (tests/fuzz_gpu.py; seeds disjoint from rounds 1-2)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3_fuzz; mkdir -p $O
timeout 1500 python -m tests.fuzz_gpu --first 100000 --cases 4000 > $O/networks.txt 2>&1; tail -2 $O/networks.txt
timeout 900 python -m tests.fuzz_gpu --subsets --first 110000 --cases 3000 > $O/subsets.txt 2>&1; tail -2 $O/subsets.txt
FW_DH_SPEC=8 FW_DH_SPEC0=4 FW_DH_SPEC1=4 FW_DH_SPEC_BELOW=100000000000 FW_DH_SPEC0_BELOW=100000000000 FW_DH_SPEC0_JOBS=100000 FW_DH_CHAINS=2 FW_DH_CHAIN_MIN=4 FW_DH_BATCH=3 timeout 900 python -m tests.fuzz_gpu --first 120000 --cases 1500 > $O/networks_lookahead_forced.txt 2>&1; tail -2 $O/networks_lookahead_forced.txt
FW_MI_ROW4=2 FW_DEV_MIN_TARGETS=8 FW_MI_SEQ=2 FW_MI_WIN0=8 FW_MI_CHUNK_MIN=1 timeout 900 python -m tests.fuzz_gpu --first 130000 --cases 1500 > $O/networks_row4_boards_forced.txt 2>&1; tail -2 $O/networks_row4_boards_forced.txt
