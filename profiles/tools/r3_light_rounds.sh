#!/bin/bash
# cfg3: deeper interleaving look-ahead in LIGHT launches only (FW_DH_SPEC0_LIGHT candidates while the last launch held fewer than FW_DH_SPEC0_LIGHT_BELOW ranks)
cd $GRAFT_REPO_ROOT
export FW_KNOBS=1
run() { echo -n "$* : "; env "$@" python bench.py $SW --steps 5 --warmup 1 --no-cpu-baseline --no-other-schedule --no-one-chain 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1), d['edges'], d['tests_per_step']['conditional_evaluated'], d['kernel_launches_per_step'])"; }
for SW in "" "--simulate-world 8 --simulate-rank 6"; do
echo "== $SW"
run A=0
for below in 100000 400000 2000000; do for d in 4 8; do
run FW_DH_SPEC0_LIGHT=$d FW_DH_SPEC0_LIGHT_BELOW=$below
done; done
done
