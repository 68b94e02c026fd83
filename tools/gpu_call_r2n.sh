#!/bin/bash
# round 2, re-entry session: K1 word-cache A/B, whole GPU suite (incl. the 2-rank data-parallel step on the native executor),
# driver-style bench without the secondary lines
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
for wc in 1 0; do
  echo "XR_K1_WORD_CACHE=$wc"; XR_K1_WORD_CACHE=$wc timeout 120 python tools/microbench_k1.py 2>&1 | tail -1
done | tee $O/r2n_k1_word_cache.txt
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 | tee $O/r2n_pytest_gpu.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-mip --no-kilo --no-unbounded --no-cpu-baseline > $O/r2n_bench_20_5.json 2> $O/r2n_bench_err.txt; tail -c 300 $O/r2n_bench_err.txt
python - <<'PY'
import json, os
p = os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'gpurun_out', 'r2n_bench_20_5.json')
try:
    d = json.loads(open(p).read().strip().splitlines()[-1])
    print('value %.3e rays/s  ms/step %.3f  normal %.3f refresh %s' % (d['value'], d['ms_per_step'], d['config']['device_ms_normal_iteration'], d['config']['device_ms_refresh_iteration']))
    for k, v in d['roofline_kernels'].items():
        print('  %-22s %8.1f us  frac %.3f  (%s)' % (k, v['avg_launch_us'], v['frac'], v['bound']))
    print('render', d.get('render_ms_per_800x800_frame'), 'f16', {a: b for a, b in d.get('ngp_f16_mlp_mode', {}).items() if a in ('value', 'ms_per_step')})
except Exception as e:
    print('bench parse failed', e, open(p).read()[-600:])
PY
