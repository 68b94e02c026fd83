#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 300 python -m pytest tests -x -q -m gpu 2>&1 | tail -30 > $O/full_gpu_suite.txt; tail -4 $O/full_gpu_suite.txt
timeout 120 python __graft_entry__.py smoke 2>&1 | tail -2 | tee $O/smoke.txt
timeout 400 python bench.py 2> $O/bench_err.txt | tail -1 > $O/bench.json; cut -c1-200 $O/bench.json; tail -2 $O/bench_err.txt
python - <<'PY'
import json, os
d = json.load(open(os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'gpurun_out', 'bench.json')))
print({k: d[k] for k in ('value', 'ms_per_step', 'render_ms_per_800x800_frame')}, d['roofline']['frac'])
m = d['mipnerf_config3']; print('mip', m['value'], m['ms_per_step'], m['cpu_baseline']['value'])
k = d['kilonerf_config5']; print('kilo', k['value'], k['cpu_baseline']['value'])
PY
