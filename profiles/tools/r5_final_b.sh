#!/bin/bash
# r05 final collection B: rocprofv3 kernel traces + PMC passes (own passes, kernel trace only) of cfg3 (headline and one chain), cfg4, cfg2, cfg3he
export ROUND=r05
for cfg in cfg3 cfg4 cfg2 cfg3he; do
  bash profiles/tools/collect_profile.sh $cfg > gpurun_out/collect_$cfg.log 2>&1
  ls gpurun_out/prof_r05_$cfg | tr '\n' ' '; echo
done
# the one-chain pass the bench line's roofline is measured on
R=$PWD; O=$R/gpurun_out/prof_r05_cfg3_one_chain; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof_one
FW_KNOBS=1 FW_DH_CHAINS=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_one -- python $R/bench.py --config cfg3 --no-cpu-baseline --no-other-schedule --no-one-chain --steps 3 --warmup 1 > $O/bench_under_rocprof.json 2>/tmp/prof_one.err
find /tmp/prof_one -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats.csv \;
grep -E "fz_subsets_seg|dh_step|dh_plan|dh_fill|gemm" $O/kernel_stats.csv | awk -F'",' '{print substr($1,1,60), $2,$3,$4}' | cut -c1-150
