#!/bin/bash
# round 2, re-entry session: the linear kernels on split bf16 operands -- parity tests, microbench against the fp32-MFMA kernel,
# the Mip-NeRF training step with both
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_linear.py tests/test_gpu_mip.py -m gpu -q -x 2>&1 | tail -4 | tee $O/r2q_pytest.txt
timeout 200 python tools/microbench_gemm_split.py 2>&1 | tail -10 | tee $O/r2q_gemm_split.txt
for kind in bf16x3 mfma; do
XR_GEMM_F32=$kind timeout 200 python - <<'PY' 2>&1 | tail -2 | tee -a $O/r2q_mip_step.txt
import os, sys, json
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch, bench
d = bench.mipnerf_config3(torch.device('cuda:0'), cpu_seconds=0.5)
print(os.environ['XR_GEMM_F32'], {k: v for k, v in d.items() if k in ('value', 'ms_per_step', 'unit')})
PY
done
