#!/bin/bash
# r05 final collection C: cfg5 at full size (one pass), N-rank replays of cfg3 / cfg4 on one GPU (with the modelled exchange cost)
O=gpurun_out/r5_final_c; mkdir -p $O
timeout 900 python bench.py --config cfg5 --steps 1 --warmup 0 --no-cpu-baseline --no-other-schedule --no-one-chain 2>$O/err_cfg5.txt | tail -1 > $O/bench_cfg5.json
python -c "
import json; l=json.loads(open('gpurun_out/r5_final_c/bench_cfg5.json').read()); print('cfg5 s %.2f edges %d'%(l['ms_per_step']/1e3, l['edges']), l['tests_per_step'])"
bash profiles/tools/simulate_world.sh cfg3 > $O/simulate_world_cfg3.txt 2>&1; cat $O/simulate_world_cfg3.txt
bash profiles/tools/simulate_world.sh cfg4 > $O/simulate_world_cfg4.txt 2>&1; cat $O/simulate_world_cfg4.txt
