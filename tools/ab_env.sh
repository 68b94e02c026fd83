#!/bin/bash
# A/B of one process-wide switch on the headline: bash tools/ab_env.sh VAR val1 val2 ... (the value `unset` leaves the variable unset)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
V=$1; shift
for val in "$@"; do
  if [ "$val" = unset ]; then unset $V; else export $V=$val; fi
  python bench.py --steps 64 --warmup 5 --no-cpu-baseline --no-mip --no-kilo --no-unbounded --no-f16 --no-strict --no-extra --no-render > /tmp/ab.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1])
k=d['roofline_kernels']
print('%-24s %.4f ms/step  %.3e rays/s  normal %.4f refresh %.4f |' % ('$V=$val', d['ms_per_step'], d['value'], d['config']['device_ms_normal_iteration'], d['config']['device_ms_refresh_iteration'] or 0), ' '.join('%s %.0f' % (n[3:], v['avg_launch_us']) for n, v in k.items()))
PY
done
