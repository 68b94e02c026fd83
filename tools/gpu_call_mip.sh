#!/bin/bash
# one GPU-box call: Mip-NeRF parity tests, stage microbench, smoke, bench (logs under gpurun_out/)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 240 python -m pytest tests/test_gpu_mip.py -x -q 2>&1 | tail -40 > $O/mip_pytest.txt; tail -5 $O/mip_pytest.txt
timeout 90 python tools/microbench_mip.py > $O/mip_microbench.txt 2>&1; cat $O/mip_microbench.txt | tail -12
timeout 120 python __graft_entry__.py smoke 2>&1 | tail -3 | tee $O/smoke.txt
timeout 300 python bench.py 2> $O/bench_err.txt | tail -1 > $O/bench.json; cut -c1-400 $O/bench.json; tail -3 $O/bench_err.txt
