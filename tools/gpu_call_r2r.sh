#!/bin/bash
# round 2, final call of the re-entry session: whole GPU suite, smoke, the linear kernels' three modes, driver-style bench
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 400 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 | tee $O/r2r_pytest_gpu.txt
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/r2r_smoke.txt
timeout 100 python tools/microbench_gemm_split.py 2>&1 | tail -6 | tee $O/r2r_gemm_split.txt
XR_GEMM_F32=bf16x3all timeout 100 python - <<'PY' 2>&1 | tail -1 | tee $O/r2r_mip_step_all.txt
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch, bench
d = bench.mipnerf_config3(torch.device('cuda:0'), cpu_seconds=0.5)
print(os.environ['XR_GEMM_F32'], {k: v for k, v in d.items() if k in ('value', 'ms_per_step', 'unit')})
PY
timeout 300 python bench.py --steps 20 --warmup 5 --no-kilo --no-unbounded --no-cpu-baseline --no-f16 > $O/r2r_bench.json 2> $O/r2r_bench_err.txt; tail -c 200 $O/r2r_bench_err.txt
python - <<'PY'
import json, os
p = os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'gpurun_out', 'r2r_bench.json')
try:
    d = json.loads(open(p).read().strip().splitlines()[-1])
    print('value %.3e rays/s  ms/step %.3f  normal %.3f refresh %s  render %.2f ms' % (d['value'], d['ms_per_step'], d['config']['device_ms_normal_iteration'], d['config']['device_ms_refresh_iteration'], d.get('render_ms_per_800x800_frame', 0)))
    for k, v in d['roofline_kernels'].items():
        print('  %-22s %8.1f us  frac %.3f  (%s)' % (k, v['avg_launch_us'], v['frac'], v['bound']))
    for k, v in d.get('roofline_kernels_render', {}).items():
        print('  render %-15s %8.1f us  frac %.3f  (%s)' % (k, v['avg_launch_us'], v['frac'], v['bound']))
    m = d.get('mipnerf_config3', {}); print('mip', {k: m.get(k) for k in ('value', 'ms_per_step', 'error')})
except Exception as e:
    print('bench parse failed', e, open(p).read()[-600:])
PY
