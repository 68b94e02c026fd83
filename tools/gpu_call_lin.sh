#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_linear.py tests/test_gpu_mip.py -x -q 2>&1 | tail -30 > $O/lin_pytest.txt; tail -6 $O/lin_pytest.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/tools/profile_mip_step.py 12 > /tmp/b.log 2>&1; tail -2 /tmp/b.log
cp $(ls /tmp/prof/*/*kernel_stats.csv | head -1) $O/mip_step_kernel_stats.csv; python $R/tools/kstats.py $O/mip_step_kernel_stats.csv | head -14
