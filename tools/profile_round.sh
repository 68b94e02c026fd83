#!/bin/bash
# End-of-round measurement on the GPU box: tests, smoke, bench, rocprofv3 kernel stats and PMC traffic of the
# bench command.  Outputs under gpurun_out/ (copy what is to be judged into profiles/).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
python -m pytest tests -m gpu -q -x 2>&1 | tail -3 | tee $O/pytest_gpu.txt
python __graft_entry__.py smoke 2>&1 | tail -2 | tee $O/smoke.txt
python bench.py 2>&1 | tail -1 > $O/bench.json; cut -c1-300 $O/bench.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py > /tmp/b.log 2>&1
cp $(ls /tmp/prof/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv; python $R/tools/kstats.py $O/kernel_stats.csv | head -24
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  d=/tmp/pmc_$(echo $c | tr ' ' '_'); rm -rf $d
  rocprofv3 --pmc $c --output-format csv -d $d -- python $R/bench.py --steps 24 --warmup 24 --no-render > /tmp/p.log 2>&1
done
python $R/tools/pmc_traffic.py $(ls /tmp/pmc_FETCH_SIZE/*/*counter_collection.csv) $(ls /tmp/pmc_WRITE_SIZE/*/*counter_collection.csv) $(ls /tmp/pmc_TCC_HIT_sum_TCC_MISS_sum/*/*counter_collection.csv) > $O/pmc_traffic.json
python - <<'PY'
import json, os
d = json.load(open(os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'gpurun_out', 'pmc_traffic.json')))
for k in ('xr_hashgrid_bwd', 'xr_hashgrid_fwd', 'xr_nerf_mlp_bwd', 'xr_nerf_mlp_fwd', 'xr_adam_step', 'k_scatter_bin', 'k_scatter_accum', 'k_hashgrid_bwd'):
    if k in d: print(k, {a: (round(b / 1e6, 1) if isinstance(b, float) and b > 1e4 else b) for a, b in d[k].items()})
PY
