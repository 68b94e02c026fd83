#!/bin/bash
# The measurement set DESIGN.md section 0 / 5 quote (round 6): gpurun --timeout 2400 -- 'bash tools/r06_final_set.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
T=r06fin
bash tools/gpu_call.sh $T tests smoke
timeout 1500 python bench.py > $O/${T}_bench_driver_style.json 2> $O/${T}_bench_driver_style_err.txt; python tools/benchsum.py $O/${T}_bench_driver_style.json
timeout 1200 python bench.py --steps 20 --warmup 5 --no-mip --no-kilo --no-unbounded --no-f16 --no-strict --no-extra --no-render > $O/${T}_bench_20_steps.json 2>/dev/null; python tools/benchsum.py $O/${T}_bench_20_steps.json
TRACE_WIN=0.8 bash tools/gpu_call.sh $T kstats trace pmc
python tools/trace_window.py $(ls /tmp/proft/*/*kernel_trace.csv | head -1) refresh > $O/${T}_trace_refresh_iteration.txt
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/profm && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profm -- python $R/tools/profile_mip_step.py 12 > /tmp/m.log 2>&1; cp $(ls /tmp/profm/*/*kernel_stats.csv | head -1) $O/${T}_mip_step_kernel_stats.csv)
python tools/kstats.py $O/${T}_mip_step_kernel_stats.csv | head -12
