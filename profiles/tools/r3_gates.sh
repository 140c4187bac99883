#!/bin/bash
# cfg3, single GPU: gate / depth of the elimination-phase look-ahead (FW_DH_SPEC_BELOW ranks, FW_DH_SPEC members)
cd $GRAFT_REPO_ROOT
export FW_KNOBS=1
run() { echo -n "$* : "; env "$@" python bench.py $SW --steps 6 --warmup 1 --no-cpu-baseline --no-other-schedule --no-one-chain 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1), d['edges'], d['tests_per_step']['conditional_evaluated'], d['kernel_launches_per_step'])"; }
SW=""
run A=0
run FW_DH_SPEC_BELOW=20000000
run FW_DH_SPEC_BELOW=30000000
run FW_DH_SPEC_BELOW=50000000
run FW_DH_SPEC_BELOW=30000000 FW_DH_SPEC=6
run FW_DH_SPEC_BELOW=30000000 FW_DH_SPEC=8
run FW_DH_SPEC_BELOW=30000000 FW_DH_SPEC=2
run FW_DH_SPEC_BELOW=1000000000
SW="--simulate-world 8 --simulate-rank 1"
run A=0
run FW_DH_SPEC_BELOW=30000000
SW="--feed-forward 0"
run A=0
run FW_DH_SPEC_BELOW=30000000
