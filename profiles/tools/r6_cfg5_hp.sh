#!/bin/bash
# r06: the plan kernel on 256 threads (default for max_k > 3; FW_DH_PLAN_SMALL=0: 1 024 threads as before), cfg5, one box
export FW_KNOBS=1
O=gpurun_out/r6_cfg5_hp; mkdir -p $O; : > $O/ab.txt
timeout 600 python -m pytest tests/test_gpu_fz.py -m gpu -q -x -k "max_k5 or long_accepted or maxk" 2>&1 | tail -2 | tee -a $O/ab.txt
for hp in 0 1; do export FW_DH_PLAN_SMALL=$hp
  timeout 900 python bench.py --config cfg5 --steps 1 --warmup 0 --no-cpu-baseline --no-other-schedule --no-one-chain 2>$O/err_$hp.txt | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('cfg5 FW_DH_PLAN_SMALL=$hp', round(l['ms_per_step'],1), l['edges'], l['network_sha256'][:12], 'kernel s', round(l['roofline']['kernel_seconds_per_step'],2))" | tee -a $O/ab.txt
done
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof_c5
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5 -- python $GRAFT_REPO_ROOT/bench.py --config cfg5 --steps 1 --warmup 0 --no-cpu-baseline --no-other-schedule --no-one-chain > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2>/dev/null
find /tmp/prof_c5 -name '*kernel_stats.csv' -exec cp {} $GRAFT_REPO_ROOT/$O/kernel_stats.csv \;
grep -E "fz_subsets_seg|dh_step|dh_plan|dh_fill" $GRAFT_REPO_ROOT/$O/kernel_stats.csv | awk -F'",' '{print substr($1,1,60), $2,$3,$4}' | cut -c1-150 | tee -a $GRAFT_REPO_ROOT/$O/ab.txt
