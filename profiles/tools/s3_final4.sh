O=gpurun_out/s3_final; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo rc=$? >> $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo rc=$? >> $O/smoke.txt
bash profiles/tools/collect_r02.sh cfg3 > $O/collect_cfg3.log 2>&1
timeout 900 python bench.py --steps 10 --warmup 1 --host-seam > $O/bench_cfg3_n1.json 2> $O/bench_cfg3_n1.err
FW_DH_CHAINS=1 timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_cfg3_onechain.json 2> /dev/null
tail -3 $O/pytest_gpu.txt; tail -2 $O/smoke.txt
