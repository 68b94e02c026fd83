#!/bin/bash
# rocprofv3 kernel stats of the scatter micro-benchmark for one switch setting: bash tools/prof_scatter.sh TAG VAR=VAL ...
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
env XR_CHILD=1 XR_QUICK=1 XR_ONLY=1 "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- python $R/tools/microbench_scatter3.py > /tmp/ps_$TAG.log 2>&1
tail -5 /tmp/ps_$TAG.log
cp $(ls /tmp/prof_$TAG/*/*kernel_stats.csv | head -1) $O/${TAG}_scatter_kernel_stats.csv
python $R/tools/kstats.py $O/${TAG}_scatter_kernel_stats.csv | grep -v "at::\|elementwise\|vectorized" | head -14
