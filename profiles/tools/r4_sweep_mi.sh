#!/bin/bash
# r04: sweeps of the persistent discrete kernel's knobs at cfg4 on the final kernels: one line per setting (ms with / without feed-forward)
export FW_KNOBS=1
run() { local name=$1; shift
  env "$@" timeout 300 python bench.py --config cfg4 --steps 8 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print(\"$name\", round(l[\"ms_per_step\"],2), round(l[\"other_schedule\"][\"ms_per_step\"],2), l[\"edges\"], \"%.4g\"%l[\"tests_per_step\"][\"conditional_evaluated\"])"
}
run default FW_X=0
for v in 16 96; do run "seq=$v" FW_MI_SEQ=$v; done
for v in 32 512; do run "win0=$v" FW_MI_WIN0=$v; done
for v in 24 96; do run "heavy=$v" FW_MI_HEAVY=$v; done
for v in 2 8; do run "seq_tail=$v" FW_MI_SEQ_TAIL=$v; done
for v in 4 16; do run "chunk_min=$v" FW_MI_CHUNK_MIN=$v; done
for v in 256 4096; do run "win0_tail=$v" FW_MI_WIN0_TAIL=$v; done
for v in 0 2; do run "ahead=$v" FW_MI_AHEAD=$v; done
for v in 1 3; do run "chains_disc=$v" FW_DH_CHAINS_DISC=$v; done
run default2 FW_X=0
