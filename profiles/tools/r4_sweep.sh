#!/bin/bash
# r04: sweeps of the device-round knobs at cfg3 on the final kernels (first window 4 096): one line per setting
export FW_KNOBS=1
run() { # name, env assignments...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-one-chain 2>/dev/null | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print(\"$name\", round(l[\"ms_per_step\"],2), round(l[\"other_schedule\"][\"ms_per_step\"],2), l[\"edges\"], l[\"kernel_launches_per_step\"], \"%.4g\"%l[\"tests_per_step\"][\"conditional_evaluated\"])"
}
run default FW_X=0
for v in 1 3 4; do run "chains=$v" FW_DH_CHAINS=$v; done
for v in 4096 65536; do run "w0_big=$v" FW_W0_BIG=$v; done
for v in 64 1024; do run "growth_small=$v" FW_DH_GROWTH_SMALL=$v; done
for v in 2 8; do run "growth_busy=$v" FW_DH_GROWTH_BUSY=$v; done
for v in 1024 8192; do run "busy_jobs=$v" FW_DH_BUSY_JOBS=$v; done
for v in 12000000 60000000; do run "spec_below=$v" FW_DH_SPEC_BELOW=$v; done
for v in 6000000 24000000; do run "spec0_below=$v" FW_DH_SPEC0_BELOW=$v; done
for v in 2048 8192; do run "spec0_jobs=$v" FW_DH_SPEC0_JOBS=$v; done
for v in 1048576 16777216; do run "small_launch=$v" FW_SMALL_LAUNCH=$v; done
for v in 0 1 4; do run "spec1=$v" FW_DH_SPEC1=$v; done
