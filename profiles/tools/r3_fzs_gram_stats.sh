cd $GRAFT_REPO_ROOT; R=$PWD; export FW_KNOBS=1 TMPDIR=/tmp
cd /tmp; rm -rf /tmp/gstats
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gstats -- python $R/profiles/tools/fzs_micro.py 40 2000 > /dev/null 2>&1
find /tmp/gstats -name '*kernel_stats.csv' -exec head -4 {} \; | cut -c1-200
rm -rf /tmp/gstats
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gstats -- python $R/profiles/tools/fzs_micro.py 100 200 > /dev/null 2>&1
find /tmp/gstats -name '*kernel_stats.csv' -exec head -4 {} \; | cut -c1-200
