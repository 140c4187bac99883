#!/bin/bash
# r06: with the 256-thread plan kernel in place, the small kernels on a high-priority stream again (FW_DH_HP), cfg5 and cfg3, one box
export FW_KNOBS=1
O=gpurun_out/r6_cfg5_hp2; mkdir -p $O; : > $O/ab.txt
for hp in 0 1; do
  FW_DH_HP=$hp timeout 900 python bench.py --config cfg5 --steps 1 --warmup 0 --no-cpu-baseline --no-other-schedule --no-one-chain 2>$O/err_$hp.txt | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('cfg5 FW_DH_HP=$hp', round(l['ms_per_step'],1), l['edges'], l['network_sha256'][:12])" | tee -a $O/ab.txt
done
for i in 1 2; do for hp in 0 1; do
  FW_DH_HP=$hp timeout 400 python bench.py --config cfg3 --steps 8 --warmup 2 --no-cpu-baseline --no-one-chain 2>/dev/null | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('cfg3 FW_DH_HP=$hp', round(l['ms_per_step'],2), round(l['other_schedule']['ms_per_step'],2), l['edges'], l['network_sha256'][:12])" | tee -a $O/ab.txt
done; done
