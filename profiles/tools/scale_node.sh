#!/bin/bash
# Day one on a multi-GPU node (r06; NO such node has been available to the builder or the driver so far: RCCL has only run with a world of one).
# Runs bench.py at N = 1 / 2 / 4 / 8 ranks (one process per GPU, torch.distributed.run, 127.0.0.1) for cfg3 / cfg4 / cfg5 in both exchange forms --
# the library's own RCCL communicator (--library-rccl: fw_comm_init / fw_level0_comm / fw_learn_network_comm) and the callback form
# (torch.distributed collectives on device buffers the library packs) -- checks that the learned network is byte-identical across N and
# across the two forms (bench.py's network_sha256), and prints the curve (ms per pass, speed-up over N = 1).
# usage: profiles/tools/scale_node.sh [--dry-run] [configs ...]      --dry-run: N = 1 only, with --force-dist (the distributed code path on one GPU)
export HSA_ENABLE_IPC_MODE_LEGACY=0
DRY=0; [ "$1" = "--dry-run" ] && { DRY=1; shift; }
CFGS=${@:-cfg3 cfg4 cfg5}
O=gpurun_out/scale_node; mkdir -p $O; : > $O/curve.txt
NG=$(python -c "import torch; print(torch.cuda.device_count())")
NS="1 2 4 8"; [ $DRY = 1 ] && NS="1"
port=29611
for cfg in $CFGS; do
  steps=6; [ $cfg = cfg5 ] && steps=1
  for form in "--library-rccl" ""; do
    for N in $NS; do
      [ $N -gt $NG ] && continue
      tag=${cfg}_$( [ -n "$form" ] && echo lib || echo cb )_n$N
      extra="--no-cpu-baseline --no-other-schedule --no-one-chain --steps $steps --warmup 1 --config $cfg $form"
      if [ $N = 1 ] && [ $DRY = 0 ]; then
        timeout 1800 python bench.py --gpus 1 $extra > $O/$tag.json 2> $O/$tag.err
      else
        port=$((port + 1))
        timeout 1800 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $port bench.py --gpus $N --force-dist $extra > $O/$tag.json 2> $O/$tag.err
      fi
      tail -1 $O/$tag.json | python -c "
import sys, json
try:
    l = json.loads(sys.stdin.read())
    print('$cfg', '$( [ -n "$form" ] && echo library-rccl || echo callback )', 'N', l['n_gpus'], 'ms', round(l['ms_per_step'], 2), 'edges', l['edges'], 'sha', l['network_sha256'][:16], 'exchange', json.dumps(l.get('exchange'))[:120])
except Exception as e:
    print('$tag FAILED', e)
" | tee -a $O/curve.txt
    done
  done
done
python - <<'PY'
import collections, re
rows = [l.split() for l in open("gpurun_out/scale_node/curve.txt") if " N " in l]
by = collections.defaultdict(list)
for r in rows:
    by[r[0]].append((r[1], int(r[3]), float(r[5]), r[9]))
for cfg, v in by.items():
    shas = {x[3] for x in v}
    print(cfg, "network identical across N and exchange forms:", len(shas) == 1, shas if len(shas) > 1 else "")
    for form in ("library-rccl", "callback"):
        t = {n: ms for f, n, ms, _ in v if f == form}
        if 1 in t:
            print("  %-12s" % form, "  ".join("N=%d %.1f ms (x%.2f)" % (n, t[n], t[1] / t[n]) for n in sorted(t)))
PY
