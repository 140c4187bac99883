cd $GRAFT_REPO_ROOT; export FW_KNOBS=1
run() { name=$1; shift; env "$@" python bench.py --p 3000 --steps 1 --warmup 0 --no-cpu-baseline --no-other-schedule --no-one-chain $EXTRA 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['edges'], d['tests_per_step'], round(d['ms_per_step'],1))"; }
EXTRA="" run recursive_dev A=1
EXTRA="" run recursive_hostpool FW_HOST_HITON=1
EXTRA="--stream-columns" run stream_gram A=1
EXTRA="--stream-columns" run stream_nogram FW_FZS_GRAM=0
EXTRA="--feed-forward 0" run recursive_dev_ff0 A=1
EXTRA="--stream-columns --feed-forward 0" run stream_gram_ff0 A=1
