
S=profiles/tools/r6_determinism.sh
C2="--p 1000 --n 500 --seed 20260930 --kind mi"
FW_LIB_PATH=$PWD/flashweave.jl_amd/libfw_r05.so $S r05lib_cfg2_ff1 4000 60 $C2 --feed-forward 1
$S fix_cfg2_ff1 10000 170 $C2 --feed-forward 1
$S fix_cfg2_ff0 10000 170 $C2 --feed-forward 0
$S fix_cfg2_ff1_r256 4000 80 $C2 --feed-forward 1 --round-size 256
cat gpurun_out/r6_determinism/summary.txt
