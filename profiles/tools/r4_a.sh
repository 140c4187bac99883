#!/bin/bash
# r04 run A: (1) does PC sampling work on this box?  cfg2 then cfg4 (the persistent discrete kernel); (2) one-chain kernel trace of the
# cfg3 headline (the pass bench.py takes its roofline from); (3) the whole GPU test suite on the housekeeping commit.
set -u
cd "${GRAFT_REPO_ROOT:-.}"; ROOT=$PWD; export TMPDIR=/tmp
O=$ROOT/gpurun_out/r4_a; rm -rf $O; mkdir -p $O
cd /tmp
rocprofv3-avail list --pc-sampling > $O/pcs_avail.txt 2>&1
rocprofv3-avail info --pc-sampling >> $O/pcs_avail.txt 2>&1
B="python $ROOT/bench.py --no-cpu-baseline --no-other-schedule --no-one-chain --steps 1 --warmup 1"
for M in stochastic host_trap; do
  if [ $M = stochastic ]; then U="--pc-sampling-unit cycles --pc-sampling-interval 1048576"; else U="--pc-sampling-unit time --pc-sampling-interval 1000"; fi
  rm -rf /tmp/pcs_$M
  timeout 300 rocprofv3 --kernel-trace --pc-sampling-beta-enabled --pc-sampling-method $M $U --output-format csv -d /tmp/pcs_$M -- $B --config cfg2 > $O/pcs_cfg2_$M.out 2> $O/pcs_cfg2_$M.err
  echo "cfg2 $M rc=$?" >> $O/log.txt
  find /tmp/pcs_$M -type f | head -20 >> $O/log.txt
  for f in $(find /tmp/pcs_$M -name '*pc_sampling*.csv'); do head -5 $f >> $O/pcs_cfg2_${M}_head.txt; wc -l $f >> $O/log.txt; done
  python $ROOT/profiles/tools/pcsamp_sum.py /tmp/pcs_$M dh_mi_target > $O/pcs_cfg2_$M.json 2>> $O/log.txt
done
# the method that produced samples: cfg4 headline
for M in stochastic host_trap; do
  n=$(python -c "import json;print(json.load(open('$O/pcs_cfg2_$M.json'))['samples'])" 2>/dev/null || echo 0)
  if [ "${n:-0}" -gt 100 ]; then
    if [ $M = stochastic ]; then U="--pc-sampling-unit cycles --pc-sampling-interval 1048576"; else U="--pc-sampling-unit time --pc-sampling-interval 1000"; fi
    rm -rf /tmp/pcs4_$M
    timeout 600 rocprofv3 --kernel-trace --pc-sampling-beta-enabled --pc-sampling-method $M $U --output-format csv -d /tmp/pcs4_$M -- $B --config cfg4 > $O/pcs_cfg4_$M.out 2> $O/pcs_cfg4_$M.err
    echo "cfg4 $M rc=$?" >> $O/log.txt
    python $ROOT/profiles/tools/pcsamp_sum.py /tmp/pcs4_$M dh_mi_target > $O/pcs_cfg4_$M.json 2>> $O/log.txt
  fi
done
# (2) cfg3, one chain: kernel trace of the pass the roofline is measured on
rm -rf /tmp/oc
FW_KNOBS=1 FW_DH_CHAINS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/oc -- python $ROOT/bench.py --no-cpu-baseline --no-other-schedule --no-one-chain --steps 3 --warmup 1 > $O/cfg3_one_chain_bench_under_rocprof.json 2> $O/oc.err
find /tmp/oc -name '*kernel_stats.csv' -exec cp {} $O/cfg3_one_chain_kernel_stats.csv \;
cd $ROOT
# (3) tests
timeout 3000 python -m pytest tests -m gpu -q -x --durations=15 > $O/pytest_gpu.txt 2>&1
tail -25 $O/pytest_gpu.txt; cat $O/log.txt; cat $O/pcs_avail.txt | head -40
