#!/bin/bash
# r05 run M: host-side trace of one cfg4 pass (where the ~6 ms outside level 0 / the conditional stage go)
O=gpurun_out/r5_m; mkdir -p $O
FW_KNOBS=1 FW_TRACE_HOST=1 timeout 600 python bench.py --config cfg4 --steps 2 --warmup 1 --no-cpu-baseline --no-other-schedule --no-one-chain 2> $O/cfg4_trace.txt > $O/bench.json
grep -v "finished at\|test routine\|state machine" $O/cfg4_trace.txt | tail -60 | cut -c1-300
