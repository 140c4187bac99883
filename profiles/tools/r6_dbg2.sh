export FW_KNOBS=1
for v in a b; do for tm in 16 0; do echo "== screen $( [ $v = a ] && echo on || echo off ), FW_FZ_TMAT=$tm"; FW_FZ_TMAT=$tm FW_LIB_PATH=$PWD/flashweave.jl_amd/libfw_dbg2$v.so timeout 600 python bench.py --config cfg3 --steps 1 --warmup 0 --no-cpu-baseline --no-other-schedule --no-one-chain 2>&1 >/dev/null | grep "fast loop\|segments (table" | tail -2; done; done
