#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 120 python __graft_entry__.py smoke 2>&1 | tail -3 | tee $O/smoke.txt
timeout 400 python bench.py 2> $O/bench_err.txt | tail -1 > $O/bench.json; cut -c1-300 $O/bench.json; tail -3 $O/bench_err.txt
python - <<'PY'
import json, os
d = json.load(open(os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'gpurun_out', 'bench.json')))
print(json.dumps(d.get('kilonerf_config5'), indent=1)[:1500])
print({k: d['mipnerf_config3'][k] for k in ('value', 'ms_per_step')}, d['mipnerf_config3']['cpu_baseline']['value'])
PY
