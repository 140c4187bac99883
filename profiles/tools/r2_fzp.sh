#!/bin/bash
# persistent Fisher-z kernel: parity tests, then cfg3 with both schedules against the level-synchronous rounds
mkdir -p gpurun_out/fzp
timeout 600 python -m pytest tests/test_gpu_fz.py -x -q -m gpu > gpurun_out/fzp/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/fzp/tests.log
tail -5 gpurun_out/fzp/tests.log
for ff in 0 1; do
  FW_TRACE_HOST=1 timeout 300 python bench.py --config cfg3 --steps 3 --warmup 1 --feed-forward $ff --no-other-schedule --no-cpu-baseline > gpurun_out/fzp/cfg3_ff$ff.json 2> gpurun_out/fzp/cfg3_ff$ff.err; echo "ff=$ff rc=$?"
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/fzp/cfg3_ff$ff.json").read().strip().splitlines()[-1])
    print("ff=$ff", d["ms_per_step"], d["edges"], d["stage_seconds_rank0"])
except Exception as e:
    print("no json", e)
PY
  grep -E "boards|watchdog|chain 0: tests" gpurun_out/fzp/cfg3_ff$ff.err | tail -4
done
FW_FZ_ROUNDS=1 timeout 300 python bench.py --config cfg3 --steps 3 --warmup 1 --feed-forward 0 --no-other-schedule --no-cpu-baseline > gpurun_out/fzp/cfg3_rounds_ff0.json 2>/dev/null
python - <<PY
import json
d = json.loads(open("gpurun_out/fzp/cfg3_rounds_ff0.json").read().strip().splitlines()[-1])
print("rounds ff=0", d["ms_per_step"], d["edges"], d["stage_seconds_rank0"])
PY
