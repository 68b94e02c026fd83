#!/bin/bash
# where should the next batch's march start inside the step?  bash tools/prefetch_point.sh
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for p in none xr_hashgrid_fwd xr_nerf_mlp_fwd xr_composite_train xr_nerf_mlp_bwd; do
  XRNERF_PREFETCH_AFTER=$p python bench.py --steps 64 --warmup 5 --no-cpu-baseline --no-mip --no-kilo --no-unbounded --no-f16 --no-strict --no-extra --no-render > /tmp/pp.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open('/tmp/pp.json').read().strip().splitlines()[-1])
k=d['roofline_kernels']
print('%-20s %.4f ms/step  %.3e rays/s  normal %.4f refresh %.4f |' % ('$p', d['ms_per_step'], d['value'], d['config']['device_ms_normal_iteration'], d['config']['device_ms_refresh_iteration'] or 0), ' '.join('%s %.0f' % (n[3:], v['avg_launch_us']) for n, v in k.items()))
PY
done
