#!/bin/bash
# kernel sequence of one grid-refresh iteration and one normal iteration (rocprofv3 kernel trace of the bench)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-mip --no-kilo --no-unbounded --no-render --no-f16 > /tmp/b.log 2>&1; tail -c 200 /tmp/b.log
T=$(ls /tmp/prof/*/*kernel_trace.csv | head -1)
python $R/tools/trace_update_iter.py $T | tee $O/r2j_refresh_iteration.txt
echo; python $R/tools/trace_window.py $T -3 | tee $O/r2j_normal_iteration.txt
