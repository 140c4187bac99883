cd $GRAFT_REPO_ROOT
FW_KNOBS=1 FW_TRACE_HOST=1 python bench.py --config cfg4 --steps 1 --warmup 1 --no-cpu-baseline --no-other-schedule --no-one-chain 2> gpurun_out/cfg4_trace.txt | cut -c1-300
grep -c . gpurun_out/cfg4_trace.txt; tail -40 gpurun_out/cfg4_trace.txt
