#!/bin/bash
# r06: the headline kernel's time follows the register allocator / scheduler more than the source: the same source under different
# LLVM scheduling / allocation options (fw_fz.hip only; libs built as libfw_v_<name>.so), one box, cfg3, one-chain kernel time + headline
export FW_KNOBS=1
O=gpurun_out/r6_flags; mkdir -p $O; : > $O/flags.txt
run() { lib=$1; FW_LIB_PATH=$PWD/flashweave.jl_amd/$lib timeout 400 python bench.py --config cfg3 --steps 6 --warmup 2 --no-cpu-baseline --no-other-schedule 2>$O/err.txt | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); r=l['roofline']; print('$lib', round(l['ms_per_step'],2), l['edges'], 'kernel ms %.2f'%(1e3*r['kernel_seconds_per_step']), l['network_sha256'][:12])" | tee -a $O/flags.txt; }
for i in 1 2; do
run libflashweave_amd.so
for v in "$@"; do run libfw_v_$v.so; done
done
