# PMC counters of the micro-benchmark (separate passes, kernel-trace only)
R=${GRAFT_REPO_ROOT:-$PWD}; cd /tmp && export TMPDIR=/tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_WAVES"; do
  tag=$(echo $grp | cut -d' ' -f1); rm -rf /tmp/pm_$tag
  M=65536 timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pm_$tag -- python $R/profiles/tools/mi_micro.py 3 > /dev/null 2>&1
  python - <<PY
import csv,glob,collections
f=glob.glob('/tmp/pm_$tag/**/*counter_collection.csv',recursive=True)
acc=collections.defaultdict(float); n=collections.Counter()
for r in csv.DictReader(open(f[0])):
    if 'mi_test_batch' in r['Kernel_Name']:
        acc[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
for k,v in acc.items(): print('%-24s per launch %.4g  per test %.1f' % (k, v/n[k], v/n[k]/65536))
PY
done
