#!/bin/bash
# r05: pass-to-pass determinism of the discrete kinds (cfg2: the network of one bench run differed in 1 of 12 passes) -- this build against the round-4 library
export FW_KNOBS=1
O=gpurun_out/r5_determinism; mkdir -p $O; : > $O/det.txt
for lib in libflashweave_amd.so libfw_r04.so; do
  for ffarg in "" "--feed-forward 0"; do
    FW_LIB_PATH=$PWD/flashweave.jl_amd/$lib timeout 200 python bench.py --config cfg2 --steps 600 --warmup 2 --no-cpu-baseline --no-other-schedule --no-one-chain --check-determinism $ffarg 2>&1 >/dev/null | grep "determinism check" | sed "s/^/$lib /" | tee -a $O/det.txt
  done
done
FW_MI_HELP_JOBS=0 timeout 200 python bench.py --config cfg2 --steps 600 --warmup 2 --no-cpu-baseline --no-other-schedule --no-one-chain --check-determinism 2>&1 >/dev/null | grep "determinism check" | sed "s/^/FW_MI_HELP_JOBS=0 /" | tee -a $O/det.txt
FW_MI_SCHED=0 timeout 200 python bench.py --config cfg2 --steps 300 --warmup 2 --no-cpu-baseline --no-other-schedule --no-one-chain --check-determinism 2>&1 >/dev/null | grep "determinism check" | sed "s/^/FW_MI_SCHED=0 /" | tee -a $O/det.txt
FW_MI_TEAM_MAX=0 timeout 200 python bench.py --config cfg2 --steps 600 --warmup 2 --no-cpu-baseline --no-other-schedule --no-one-chain --check-determinism 2>&1 >/dev/null | grep "determinism check" | sed "s/^/FW_MI_TEAM_MAX=0 /" | tee -a $O/det.txt
FW_MI_ROW4=0 timeout 200 python bench.py --config cfg2 --steps 600 --warmup 2 --no-cpu-baseline --no-other-schedule --no-one-chain --check-determinism 2>&1 >/dev/null | grep "determinism check" | sed "s/^/FW_MI_ROW4=0 /" | tee -a $O/det.txt
