#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 100 python -m pytest tests/test_gpu_linear.py -x -q 2>&1 | tail -3 | tee $O/lin_pytest.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py > /tmp/b.log 2>&1; tail -c 300 /tmp/b.log
cp $(ls /tmp/prof/*/*kernel_stats.csv | head -1) $O/bench_kernel_stats.csv; python $R/tools/kstats.py $O/bench_kernel_stats.csv | head -12
