#!/bin/bash
# r06: 2 048 striding workgroups for the size-3 table kernel became the default: the cfg3 bench line again, then the fz parity tests and the cfg3 full-size tests
mkdir -p gpurun_out/r6_last2
timeout 120 python bench.py --steps 20 --warmup 1 2>/dev/null | tail -1 > gpurun_out/r6_last2/bench_cfg3.json; python -c "import json; l=json.load(open('gpurun_out/r6_last2/bench_cfg3.json')); print(l['ms_per_step'], l['value'], l['other_schedule']['ms_per_step'], l['roofline']['frac'], l['edges'], l['network_sha256'][:12])"
timeout 200 python -m pytest tests/test_gpu_fz.py tests/test_gpu_fullsize.py -q -x -m gpu -k "not cfg4 and not cfg5" 2>&1 | tail -4 | tee gpurun_out/r6_last2/pytest.txt
