#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
for v in 1 2 3 4 5 6; do timeout 60 tools/scatter_timing_v$v 2>&1 | tail -3; done | tee $O/r2e_scatter_variants.txt
