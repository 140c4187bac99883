#!/bin/bash
# r06: local correlation matrices for the max_k 4-5 kernels (FW_FZ_TMAT: 0 = the p x p matrix, default 16 = targets with at least 16 neighbours), parity tests first, then cfg5 on one box
export FW_KNOBS=1
O=gpurun_out/r6_cfg5_tmat; mkdir -p $O; : > $O/ab.txt
timeout 900 python -m pytest tests/test_gpu_fz.py tests/test_gpu_maxk.py -x -q -m gpu -k "max_k or table or golden or size_4_5" 2>&1 | tail -5 | tee $O/pytest.txt
timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "cfg5" 2>&1 | tail -5 | tee -a $O/pytest.txt
for tm in 16 0; do
  FW_TRACE_HOST=1 FW_FZ_TMAT=$tm timeout 900 python bench.py --config cfg5 --steps 1 --warmup 0 --no-cpu-baseline --no-other-schedule --no-one-chain 2>$O/err_$tm.txt | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('cfg5 FW_FZ_TMAT=$tm', round(l['ms_per_step'],1), l['edges'], l['network_sha256'][:12], 'kernel s', round(l['roofline']['kernel_seconds_per_step'],2))" | tee -a $O/ab.txt
  grep "local correlation" $O/err_$tm.txt | sort | uniq -c | sort -rn | head -5 | tee -a $O/ab.txt
done
