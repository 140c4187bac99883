#!/bin/bash
# r05: a pass of cfg2 (feed-forward, one round) differs from the others about once in 1 000 passes (one target's PC list): which feature of the
# persistent discrete kernel carries it -- 6 000 passes each
export FW_KNOBS=1
O=gpurun_out/r5_race; mkdir -p $O; : > $O/bisect.txt
run() { echo "== $*" | tee -a $O/bisect.txt; env "$@" timeout 100 python profiles/tools/determinism_mi.py 6000 cfg2 1 2>&1 | grep "DIFFERS\|^passes" | tail -4 | cut -c1-200 | tee -a $O/bisect.txt; }
run FW_X=0
run FW_MI_HELP_JOBS=0
run FW_MI_TEAM_MAX=0
run FW_MI_AHEAD=0
run FW_MI_ROW4=0
run FW_LIB_PATH=$PWD/flashweave.jl_amd/libfw_r04.so
