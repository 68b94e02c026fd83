#!/bin/bash
# round 2, final call: whole GPU suite, smoke, driver-style bench, real-pixels PSNR record, rocprofv3 kernel stats and PMC traffic passes of the bench
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 | tee $O/r2m_pytest_gpu.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $O/r2m_smoke.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/r2m_bench_20_5.json 2> $O/r2m_bench_err.txt; tail -c 400 $O/r2m_bench_err.txt; python - <<'PY'
import json, os
p = os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'gpurun_out', 'r2m_bench_20_5.json')
try:
    d = json.loads(open(p).read().strip().splitlines()[-1])
    print('value %.3e rays/s  ms/step %.3f  rays/step %.0f  samples/ray %.1f' % (d['value'], d['ms_per_step'], d['config']['rays_per_step'], d['config']['samples_per_ray']))
    print('roofline', {k: (round(v, 3) if isinstance(v, float) else v) for k, v in d['roofline'].items() if k in ('kernel', 'frac', 'avg_launch_us', 'achieved')})
    for k, v in d['roofline_kernels'].items():
        print('  %-22s %8.1f us  frac %.3f  (%s)' % (k, v['avg_launch_us'], v['frac'], v['bound']))
    for k in ('render_ms_per_800x800_frame', 'render_ms_per_800x800_frame_early_termination_1e-4', 'cpu_baseline', 'cpu_baseline_vanilla_nerf_config1'):
        print(k, d.get(k))
    for k in ('ngp_config4_unbounded', 'mipnerf_config3', 'kilonerf_config5'):
        v = d.get(k, {})
        print(k, {a: b for a, b in v.items() if a not in ('workload', 'kernels', 'cpu_baseline')})
except Exception as e:
    print('bench parse failed', e, open(p).read()[-600:])
PY
timeout 300 python tools/train_real_lego.py oracle/_ref/data/lego $O/r2m_real_lego_psnr.json 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py --steps 128 --warmup 5 --no-cpu-baseline --no-mip --no-kilo --no-unbounded --no-render > /tmp/b.log 2>&1; tail -c 300 /tmp/b.log
cp $(ls /tmp/prof/*/*kernel_stats.csv | head -1) $O/r2m_bench_kernel_stats.csv; python $R/tools/kstats.py $O/r2m_bench_kernel_stats.csv | head -24
ARGS="--steps 64 --warmup 5 --no-cpu-baseline --no-mip --no-kilo --no-unbounded --no-render --no-f16"
for c in FETCH_SIZE WRITE_SIZE; do
  d=/tmp/pmc_$c; rm -rf $d
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $d -- python $R/bench.py $ARGS > /tmp/p.log 2>&1
done
python $R/tools/pmc_traffic.py $(ls /tmp/pmc_FETCH_SIZE/*/*counter_collection.csv | head -1) $(ls /tmp/pmc_WRITE_SIZE/*/*counter_collection.csv | head -1) > $O/r2m_pmc_traffic.json
python - <<'PY'
import json, os
d = json.load(open(os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'gpurun_out', 'r2m_pmc_traffic.json')))
for k in ('xr_hashgrid_bwd', 'xr_hashgrid_fwd', 'xr_nerf_mlp_bwd', 'xr_nerf_mlp_fwd', 'xr_adam_step', 'xr_live_rows', 'xr_composite_train'):
    if k in d: print(k, {a: (round(b / 1e6, 1) if isinstance(b, float) and b > 1e4 else b) for a, b in d[k].items()})
PY
