#!/bin/bash
# round 2, call A: hash-grid variants micro-benchmark, quick GPU parity of the touched kernels, the new bench.py
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python tools/microbench_hash.py all 2>&1 | grep -v Warning | tee $O/r2a_microbench_hash.txt
timeout 300 python -m pytest tests/test_gpu_tcnn.py tests/test_gpu_fullsize.py tests/test_gpu_network.py -x -q -m gpu 2>&1 | tail -5 | tee $O/r2a_pytest.txt
timeout 400 python bench.py --steps 20 --warmup 5 --no-mip --no-kilo > $O/r2a_bench_20_5.json 2> $O/r2a_bench_err.txt; cut -c1-1500 $O/r2a_bench_20_5.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/tools/microbench_hash.py bwd > /tmp/b.log 2>&1; tail -c 600 /tmp/b.log
for f in $(ls /tmp/prof/*/*kernel_stats.csv); do python $R/tools/kstats.py $f | head -8; done 2>&1 | tee $O/r2a_bwd_kstats.txt
