#!/bin/bash
# cfg5 long-list kernel (level-3 position tables): four workgroups per CU (128 VGPRs, tables of 320 / 288 positions) against three (168 VGPRs, 448)
cd $GRAFT_REPO_ROOT
for cfg in "4 320" "4 288"; do set -- $cfg
  touch flashweave.jl_amd/csrc/fw_fz.hip
  make -C flashweave.jl_amd/csrc EXTRA="-DFW_HIGHK_OCC_LONG=$1 -DFZ_L3_CAP=$2" > gpurun_out/make_occ4.log 2>&1; grep -c error gpurun_out/make_occ4.log
  timeout 900 python bench.py --config cfg5 --steps 1 --warmup 0 --no-cpu-baseline --no-other-schedule --no-one-chain 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('OCC_LONG=$1 L3_CAP=$2', round(d['ms_per_step']/1e3,2), 's', d['edges'], 'edges')"
done
touch flashweave.jl_amd/csrc/fw_fz.hip; make -C flashweave.jl_amd/csrc > /dev/null 2>&1
