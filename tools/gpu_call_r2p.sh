#!/bin/bash
# round 2, re-entry session: split forward with dynamic tile tickets -- tests, stand-alone time, in-loop time (bench)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_tcnn.py -m gpu -q -x -k "mlp_fwd" 2>&1 | tail -4 | tee $O/r2p_pytest_fwd.txt
timeout 200 python tools/microbench_mlp_fwd_split.py 2>&1 | tail -8 | tee $O/r2p_mlp_fwd_split.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-mip --no-kilo --no-unbounded --no-cpu-baseline --no-f16 > $O/r2p_bench.json 2> $O/r2p_bench_err.txt; tail -c 200 $O/r2p_bench_err.txt
python - <<'PY'
import json, os
p = os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'gpurun_out', 'r2p_bench.json')
try:
    d = json.loads(open(p).read().strip().splitlines()[-1])
    print('value %.3e rays/s  ms/step %.3f  normal %.3f refresh %s  render %.2f ms' % (d['value'], d['ms_per_step'], d['config']['device_ms_normal_iteration'], d['config']['device_ms_refresh_iteration'], d.get('render_ms_per_800x800_frame', 0)))
    for k, v in d['roofline_kernels'].items():
        print('  %-22s %8.1f us  frac %.3f  (%s)' % (k, v['avg_launch_us'], v['frac'], v['bound']))
except Exception as e:
    print('bench parse failed', e, open(p).read()[-600:])
PY
