#!/bin/bash
# discrete persistent kernel: how long an idle wavefront sleeps between two looks at the boards (waiting for its own board / in the tail)
cd $GRAFT_REPO_ROOT
run() { name=$1; shift; cfg=$1; shift; ff=$1; shift
  timeout 300 python bench.py --config $cfg --feed-forward $ff --steps 4 --warmup 1 --no-cpu-baseline --no-other-schedule --no-one-chain 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$name', round(d['ms_per_step'],2), 'cond', round(1e3*d['stage_seconds_rank0']['conditional'],2), 'edges', d['edges'])"
}
for nap in "1 1" "4 4" "8 16" "2 32"; do set -- $nap
  touch flashweave.jl_amd/csrc/fw_devhiton.hip; make -C flashweave.jl_amd/csrc EXTRA="-DMI_NAP_WAIT=$1 -DMI_NAP_TAIL=$2" > /dev/null 2>&1
  run wait$1_tail$2_cfg4_ff1 cfg4 1
  run wait$1_tail$2_cfg4_ff0 cfg4 0
  run wait$1_tail$2_cfg2 cfg2 1
done
touch flashweave.jl_amd/csrc/fw_devhiton.hip; make -C flashweave.jl_amd/csrc > /dev/null 2>&1
