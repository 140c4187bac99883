#!/bin/bash
# r04 final collection, part B: cfg5 at full size (bench + kernel stats + PMC), variant S whole, fuzz sweeps on the final kernels, GPU test suite
set -u
cd "${GRAFT_REPO_ROOT:-.}"; ROOT=$PWD; export TMPDIR=/tmp
O=$ROOT/gpurun_out/r4_final_b; rm -rf $O; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; grep -E 'passed|failed' $O/pytest_gpu.txt
python bench.py --config cfg5 --steps 1 --warmup 1 --cpu-seconds 10 > $O/bench_cfg5_n1.json 2>/dev/null
python -c "
import json
d=json.loads(open('$O/bench_cfg5_n1.json').read().strip().splitlines()[-1]); print('cfg5 s', d['ms_per_step']/1e3, 'edges', d['edges'], 'frac', d['roofline']['frac'], d['roofline']['bound'])"
ROUND=r04 bash profiles/tools/collect_profile.sh cfg5 > $O/collect_cfg5.log 2>&1
python bench.py --stream-columns --steps 1 --warmup 1 --no-cpu-baseline --no-other-schedule > $O/bench_cfg3_stream_columns_whole.json 2>/dev/null
export FW_KNOBS=1
timeout 1500 python -m tests.fuzz_gpu --first 200000 --cases 3000 > $O/fuzz_networks.txt 2>&1; tail -1 $O/fuzz_networks.txt
timeout 900 python -m tests.fuzz_gpu --subsets --first 210000 --cases 2500 > $O/fuzz_subsets.txt 2>&1; tail -1 $O/fuzz_subsets.txt
FW_DH_SPEC=8 FW_DH_SPEC0=4 FW_DH_SPEC1=4 FW_DH_SPEC_BELOW=100000000000 FW_DH_SPEC0_BELOW=100000000000 FW_DH_SPEC0_JOBS=100000 FW_DH_CHAINS=2 FW_DH_CHAIN_MIN=4 FW_DH_BATCH=3 FW_DH_FUSE=2 timeout 900 python -m tests.fuzz_gpu --first 220000 --cases 1200 > $O/fuzz_lookahead_forced_fused_round.txt 2>&1; tail -1 $O/fuzz_lookahead_forced_fused_round.txt
FW_MI_ROW4=2 FW_DEV_MIN_TARGETS=8 FW_MI_SEQ=2 FW_MI_WIN0=8 FW_MI_CHUNK_MIN=1 timeout 900 python -m tests.fuzz_gpu --first 230000 --cases 1200 > $O/fuzz_row4_boards_forced.txt 2>&1; tail -1 $O/fuzz_row4_boards_forced.txt
