#!/bin/bash
# r06: pass-to-pass determinism of the discrete kinds under contention (two worker processes on one GPU, every pass against the first).
# usage: r6_determinism.sh <label> <passes per process> <seconds cap per process> [worker args ...]     (FW_LIB_PATH selects another library)
export FW_KNOBS=1
O=gpurun_out/r6_determinism; mkdir -p $O
label=$1; passes=$2; secs=$3; shift 3
go=$O/go_$label; rm -f $go $go.*
for w in A B; do
  timeout $((secs + 200)) python tests/determinism_worker.py --passes $passes --seconds $secs --start-file $go "$@" > $O/${label}_$w.json 2> $O/${label}_$w.err &
  eval pid_$w=$!
done
for i in $(seq 1 2000); do [ $(ls $go.* 2>/dev/null | wc -l) -ge 2 ] && break; sleep 0.05; done
touch $go
wait $pid_A; ra=$?; wait $pid_B; rb=$?
rm -f $go $go.*
for w in A B; do echo "$label $w (exit $( [ $w = A ] && echo $ra || echo $rb )): $(tail -1 $O/${label}_$w.json | cut -c1-900)"; done | tee -a $O/summary.txt
