#!/bin/bash
# r06 final collection A: pytest -m gpu, the bench lines, kernel traces + counter passes (cfg3 two chains / one chain, cfg3he), scale_node dry run, cfg5
export FW_KNOBS=1
O=gpurun_out/r6_final_a; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --durations=12 > $O/pytest_gpu.txt 2>&1; tail -18 $O/pytest_gpu.txt
export ROUND=r06
for cfg in cfg3 cfg3he; do bash profiles/tools/collect_profile.sh $cfg > gpurun_out/collect_$cfg.log 2>&1; ls gpurun_out/prof_r06_$cfg | tr '\n' ' '; echo; done
R=$PWD; P=$R/gpurun_out/prof_r06_cfg3_one_chain; mkdir -p $P
( cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof_one; FW_KNOBS=1 FW_DH_CHAINS=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_one -- python $R/bench.py --config cfg3 --no-cpu-baseline --no-other-schedule --no-one-chain --steps 3 --warmup 1 > $P/bench_under_rocprof.json 2>/tmp/prof_one.err; find /tmp/prof_one -name '*kernel_stats.csv' -exec cp {} $P/kernel_stats.csv \; )
grep -E "fz_subsets_seg|dh_step|dh_plan|dh_fill|gemm|tmat" $P/kernel_stats.csv | awk -F'",' '{print substr($1,1,60), $2,$3,$4}' | cut -c1-150
bash profiles/tools/scale_node.sh --dry-run cfg3 cfg4 2>&1 | tail -8
timeout 900 python bench.py --config cfg5 --steps 1 --warmup 0 --no-cpu-baseline --no-other-schedule --no-one-chain 2>$O/err_cfg5.txt | tail -1 > $O/bench_cfg5.json
python -c "import json; l=json.loads(open('$O/bench_cfg5.json').read()); print('cfg5 ms', l['ms_per_step'], 'edges', l['edges'], l['tests_per_step'])"
