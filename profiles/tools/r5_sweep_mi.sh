#!/bin/bash
# r05: knobs of the persistent discrete kernel at cfg4 on the final kernels of the round: one line per setting (ms with / without feed-forward)
export FW_KNOBS=1
O=gpurun_out/r5_sweep_mi; mkdir -p $O
run() { local name=$1; shift
  env "$@" timeout 300 python bench.py --config cfg4 --steps 6 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print(\"$name\", round(l[\"ms_per_step\"],2), round(l[\"other_schedule\"][\"ms_per_step\"],2), l[\"edges\"], \"%.4g\"%l[\"tests_per_step\"][\"conditional_evaluated\"], (l[\"roofline\"].get(\"level0\") or {}).get(\"frac\"))" | tee -a $O/sweep.txt
}
run default FW_X=0
for v in 32 64; do run "seq=$v" FW_MI_SEQ=$v; done
for v in 32 64; do run "heavy=$v" FW_MI_HEAVY=$v; done
for v in 64 128; do run "team_min=$v" FW_MI_TEAM_MIN=$v; done
for v in 128 256; do run "team_max=$v" FW_MI_TEAM_MAX=$v; done
for v in 2 8; do run "seq_tail=$v" FW_MI_SEQ_TAIL=$v; done
for v in 1 3; do run "chains_disc=$v" FW_DH_CHAINS_DISC=$v; done
run default2 FW_X=0
