#!/bin/bash
# r06: the new max_k 4 / 5 fuzz test alone (time), then the default bench line once more on the final library
( time timeout 600 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -k "max_k_4_5" ) 2>&1 | tail -6
timeout 600 python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_default.json; python -c "import json; l=json.load(open('gpurun_out/bench_default.json')); print(l['ms_per_step'], l['value'], l['roofline']['frac'], l['cpu_baseline'])"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
