#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_kilo.py -x -q 2>&1 | tail -40 > $O/kilo_pytest.txt; tail -6 $O/kilo_pytest.txt
timeout 120 python tools/microbench_kilo.py 2>&1 | tail -3 | tee $O/kilo_microbench.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/tools/microbench_kilo.py > /tmp/b.log 2>&1; tail -1 /tmp/b.log
cp $(ls /tmp/prof/*/*kernel_stats.csv | head -1) $O/kilo_kernel_stats.csv; python $R/tools/kstats.py $O/kilo_kernel_stats.csv | head -12
