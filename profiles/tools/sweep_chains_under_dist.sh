export FW_KNOBS=1  # the library reads FW_* knobs only when this is set
run() { echo -n "$* : "; env "$@" python bench.py --gpus 1 --force-dist --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1), round(d['other_schedule']['ms_per_step'],1), d['edges'], d['exchange']['us_per_round_rank0'])"; }
run FW_DH_CHAINS=3 GPU_MAX_HW_QUEUES=8
run FW_DH_CHAINS=3 GPU_MAX_HW_QUEUES=6
run2() { echo -n "nodist $* : "; env "$@" python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1), round(d['other_schedule']['ms_per_step'],1), d['edges'])"; }
run2 FW_DH_CHAINS=3 GPU_MAX_HW_QUEUES=8
run2 FW_DH_CHAINS=4 GPU_MAX_HW_QUEUES=8
run2 FW_DH_CHAINS=3 GPU_MAX_HW_QUEUES=2
