#!/bin/bash
# r05: window knobs of the fz device rounds at cfg3 with the jobs' stop words in place (a stop now ends the later segments of its window
# early, so larger windows cost less speculation than when r04 swept them): one line per setting
export FW_KNOBS=1
O=gpurun_out/r5_sweep_fz; mkdir -p $O
run() { local name=$1; shift
  env "$@" timeout 300 python bench.py --config cfg3 --steps 8 --warmup 2 --no-cpu-baseline --no-one-chain 2>/dev/null | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print(\"$name\", round(l[\"ms_per_step\"],2), round(l[\"other_schedule\"][\"ms_per_step\"],2), l[\"edges\"], \"%.5g\"%l[\"tests_per_step\"][\"conditional_evaluated\"])" | tee -a $O/sweep.txt
}
run default FW_X=0
for v in 4096 8192; do run "w0_small=$v" FW_W0_SMALL=$v; done
for v in 65536 131072; do run "w0_big=$v" FW_W0_BIG=$v; done
for v in 8 16; do run "growth=$v" FW_DH_GROWTH=$v; done
run "w0_small=4096,w0_big=65536" FW_W0_SMALL=4096 FW_W0_BIG=65536
run default2 FW_X=0
