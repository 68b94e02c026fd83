#!/bin/bash
# round 2, call D: gather roofline probe, forward ablations (levels, position layout), scatter ablations
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 120 tools/gather_probe 2>&1 | tee $O/r2d_gather_probe.txt
XR_FWD_ABLATE=1 XR_FWD_MODES=8,16 timeout 300 python tools/microbench_hash.py fwd 2>&1 | grep -v "amdgpu.ids" | tee $O/r2d_fwd_ablate.txt
for v in scatter_timing scatter_timing_noatomic scatter_timing_noload; do echo "== $v"; timeout 60 tools/$v 2>&1 | tail -4; done | tee $O/r2d_scatter_ablate.txt
