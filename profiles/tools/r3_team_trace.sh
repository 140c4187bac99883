#!/bin/bash
# discrete persistent kernel: where a test's time goes (build with -DFW_MI_TICKS), team rounds on / off
cd $GRAFT_REPO_ROOT
export FW_KNOBS=1
touch flashweave.jl_amd/csrc/fw_devhiton.hip; make -C flashweave.jl_amd/csrc EXTRA=-DFW_MI_TICKS > gpurun_out/make_ticks.log 2>&1
for cfg in "0 64" "64 64" "64 256" "128 256" "96 256"; do
set -- $cfg
echo "FW_MI_TEAM_MIN=$1 FW_MI_TEAM_MAX=$2"
FW_MI_TEAM_MIN=$1 FW_MI_TEAM_MAX=$2 FW_TRACE_HOST=1 python bench.py --config cfg4 --steps 1 --warmup 0 --no-cpu-baseline --no-other-schedule --no-one-chain 2>&1 >/dev/null | grep -E "finished at|run by a workgroup|boards |team rounds|test routine|device rounds \(all|conditional stage" | tail -13 | cut -c1-330
done
touch flashweave.jl_amd/csrc/fw_devhiton.hip; make -C flashweave.jl_amd/csrc > /dev/null 2>&1
