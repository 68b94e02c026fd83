#!/bin/bash
# kernel sequence of one steady-state training iteration of the bench in both MLP precisions (rocprofv3 kernel trace);
# LEGS may add f32_dense / f16_dense (XR_MLP_LIVE=0: backward over every row)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 40 --warmup 5 --no-cpu-baseline --no-mip --no-kilo --no-unbounded --no-render --no-f16"
for leg in ${LEGS:-f32 f16}; do
  case $leg in
    f32) export XRNERF_MLP_PRECISION=f32; unset XR_MLP_LIVE;;
    f16) export XRNERF_MLP_PRECISION=f16; unset XR_MLP_LIVE;;
    f32_dense) export XRNERF_MLP_PRECISION=f32; export XR_MLP_LIVE=0;;
    f16_dense) export XRNERF_MLP_PRECISION=f16; export XR_MLP_LIVE=0;;
  esac
  rm -rf /tmp/prof; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -- python $R/bench.py $ARGS > /tmp/b.log 2>&1
  echo "== $leg"; tail -1 /tmp/b.log | cut -c 1-160
  T=$(ls /tmp/prof/*/*kernel_trace.csv | head -1)
  python $R/tools/trace_window.py $T -3 | tee $O/r2k_normal_iteration_$leg.txt
done
