O=gpurun_out/s3e; mkdir -p $O
for P in 30000; do
timeout 600 python bench.py --config cfg5 --p $P --steps 1 --warmup 1 --no-cpu-baseline --no-other-schedule > $O/hk_$P.json 2> $O/hk_$P.err
FW_NO_HK=1 timeout 600 python bench.py --config cfg5 --p $P --steps 1 --warmup 1 --no-cpu-baseline --no-other-schedule > $O/gen_$P.json 2> $O/gen_$P.err
python - <<PY
import json
for t in ("hk","gen"):
    d=json.load(open("gpurun_out/s3e/%s_$P.json"%t))
    print(t, $P, d['ms_per_step'], d['edges'], d['tests_per_step'], d['roofline'].get('evaluated_tests_per_s_in_kernel'), d['stage_seconds_rank0']['conditional'])
PY
done
