#!/bin/bash
# r06 final collection C (the 256-thread plan kernel became the default after A / B): cfg3's traces and counter passes again, then every bench line again
export FW_KNOBS=1 ROUND=r06
bash profiles/tools/collect_profile.sh cfg3 > gpurun_out/collect_cfg3.log 2>&1; ls gpurun_out/prof_r06_cfg3 | tr '\n' ' '; echo
R=$PWD; P=$R/gpurun_out/prof_r06_cfg3_one_chain; mkdir -p $P
( cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof_one; FW_KNOBS=1 FW_DH_CHAINS=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_one -- python $R/bench.py --config cfg3 --no-cpu-baseline --no-other-schedule --no-one-chain --steps 3 --warmup 1 > $P/bench_under_rocprof.json 2>/tmp/prof_one.err; find /tmp/prof_one -name '*kernel_stats.csv' -exec cp {} $P/kernel_stats.csv \; )
unset FW_KNOBS
bash profiles/tools/r6_final_b.sh 2>&1 | grep -v "^\.\|passed\|durations\|call  " | tail -8
