#!/bin/bash
# cfg3: interleaving look-ahead in the light feed-forward rounds (1 024 live jobs: the default gate "fewer than 512 live jobs" keeps it off)
cd $GRAFT_REPO_ROOT
export FW_KNOBS=1
run() { echo -n "$* : "; env "$@" python bench.py $SW --steps 5 --warmup 1 --no-cpu-baseline --no-other-schedule --no-one-chain 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1), d['edges'], d['tests_per_step']['conditional_evaluated'], d['kernel_launches_per_step'])"; }
for SW in "" "--simulate-world 8 --simulate-rank 6"; do
echo "== $SW"
run A=0
run FW_DH_SPEC0_JOBS=4096
run FW_DH_SPEC0_JOBS=4096 FW_DH_SPEC0=4
run FW_DH_SPEC0_JOBS=4096 FW_DH_SPEC0=8
run FW_DH_SPEC0_JOBS=4096 FW_DH_SPEC0=4 FW_DH_SPEC0_BELOW=2000000
run FW_DH_SPEC0_JOBS=4096 FW_DH_SPEC0=8 FW_DH_SPEC0_BELOW=2000000
done
