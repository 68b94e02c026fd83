#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 | tee $O/r2h_pytest_gpu.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-mip --no-kilo --no-unbounded --no-cpu-baseline --no-f16 > $O/r2h_bench.json 2> $O/r2h_bench_err.txt; tail -c 300 $O/r2h_bench_err.txt
python - <<'PY'
import json, os
d = json.loads(open(os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'gpurun_out', 'r2h_bench.json')).read().strip().splitlines()[-1])
print('f32: %.3e rays/s  %.3f ms/step' % (d['value'], d['ms_per_step']))
for k, v in d['roofline_kernels'].items(): print('  %-22s %8.1f us  frac %.3f' % (k, v['avg_launch_us'], v['frac']))
PY
