timeout 600 python -m pytest tests/test_gpu_mi.py tests/test_gpu_dist.py -q -x -k "level0 or golden or network or sharded" 2>&1 | tail -3
FW_L0_MFMA=1 L0_ABLATE_SET="0 1 3 5 6" timeout 800 python profiles/tools/l0_ablate.py 2>&1 | grep -v amdgpu.ids
cd /tmp && export TMPDIR=/tmp
FW_L0_DBG=0 FW_KNOBS=1 FW_L0_VERBOSE=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_l -o l -- python /root/repo/profiles/tools/l0_ablate.py child 2>&1 | grep "level-0" | head -2
f=$(find /tmp/prof_l -name "*kernel_stats.csv" | head -1); ls -R /tmp/prof_l | head; python - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:6]:
    print(r['Name'][:50], r['Calls'], round(float(r['AverageNs'])/1e3,1),'us')
PY
