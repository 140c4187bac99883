for N in 1 2 4 8; do echo -n "simulate-world $N : "; python bench.py --simulate-world $N --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1), round(d['other_schedule']['ms_per_step'],1))"; done
