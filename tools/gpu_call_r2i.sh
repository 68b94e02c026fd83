#!/bin/bash
# round 2: profiles of the bench command -- rocprofv3 kernel stats, PMC traffic passes (FETCH_SIZE / WRITE_SIZE / L2 hit), then the bench line itself
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 64 --warmup 5 --no-cpu-baseline --no-mip --no-kilo --no-unbounded --no-render --no-f16"
rm -rf /tmp/prof; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py $ARGS > /tmp/b.log 2>&1; tail -c 200 /tmp/b.log
cp $(ls /tmp/prof/*/*kernel_stats.csv | head -1) $O/r2i_bench_kernel_stats.csv; python $R/tools/kstats.py $O/r2i_bench_kernel_stats.csv | head -16
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  d=/tmp/pmc_$(echo $c | tr ' ' '_'); rm -rf $d
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $d -- python $R/bench.py $ARGS > /tmp/p.log 2>&1
done
python $R/tools/pmc_traffic.py $(ls /tmp/pmc_FETCH_SIZE/*/*counter_collection.csv | head -1) $(ls /tmp/pmc_WRITE_SIZE/*/*counter_collection.csv | head -1) $(ls /tmp/pmc_TCC_HIT_sum_TCC_MISS_sum/*/*counter_collection.csv | head -1) > $O/r2i_pmc_traffic.json
python - <<'PY'
import json, os
d = json.load(open(os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'gpurun_out', 'r2i_pmc_traffic.json')))
for k in ('xr_hashgrid_bwd', 'xr_hashgrid_fwd', 'xr_nerf_mlp_bwd', 'xr_nerf_mlp_fwd', 'xr_adam_step', 'k_scatter_bin2', 'k_scatter_accum2', 'k_hashgrid_bwd', 'k_hashgrid_fwd'):
    if k in d: print(k, {a: (round(b / 1e6, 1) if isinstance(b, float) and b > 1e4 else b) for a, b in d[k].items()})
PY
