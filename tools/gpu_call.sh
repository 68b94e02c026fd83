#!/bin/bash
# One parametrised GPU-box script (replaces the per-call scripts of rounds 1-2):
#   gpurun --timeout T -- 'bash tools/gpu_call.sh TAG stage [stage ...]'
# Every stage writes under gpurun_out/ with the TAG prefix; summaries worth keeping are copied to profiles/ by hand.
# Stages: tests | smoke | bench[:args] | kstats[:args] | trace[:args] | pmc[:args] | scatter[:quick] | fwd | py:<script and args>
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
TAG=$1; shift
cd $R
for st in "$@"; do
  name=${st%%:*}; arg=""; [ "$st" != "$name" ] && arg=${st#*:}
  echo "=== stage $st"
  case $name in
    tests)   timeout 1200 python -m pytest tests -m gpu -q -x --durations=8 $arg 2>&1 | tail -30 | tee $O/${TAG}_pytest_gpu.txt ;;
    smoke)   timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee $O/${TAG}_smoke.txt ;;
    bench)   timeout 1200 python bench.py ${arg:---steps 20 --warmup 5} > $O/${TAG}_bench.json 2> $O/${TAG}_bench_err.txt; tail -c 300 $O/${TAG}_bench_err.txt
             python tools/benchsum.py $O/${TAG}_bench.json ;;
    kstats)  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- \
               python $R/bench.py ${arg:---steps 128 --warmup 5 --no-cpu-baseline --no-mip --no-kilo --no-unbounded --no-render --no-f16 --no-strict --no-extra} > /tmp/b.log 2>&1; tail -c 300 /tmp/b.log)
             cp $(ls /tmp/prof/*/*kernel_stats.csv | head -1) $O/${TAG}_bench_kernel_stats.csv; python tools/kstats.py $O/${TAG}_bench_kernel_stats.csv | head -30 ;;
    pmc)     for c in FETCH_SIZE WRITE_SIZE; do d=/tmp/pmc_$c; rm -rf $d
               (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --pmc $c --output-format csv -d $d -- \
                 python $R/bench.py ${arg:---steps 64 --warmup 5 --no-cpu-baseline --no-mip --no-kilo --no-unbounded --no-render --no-f16 --no-strict --no-extra} > /tmp/p.log 2>&1)
             done
             python tools/pmc_traffic.py $(ls /tmp/pmc_FETCH_SIZE/*/*counter_collection.csv | head -1) $(ls /tmp/pmc_WRITE_SIZE/*/*counter_collection.csv | head -1) > $O/${TAG}_pmc_traffic.json
             python tools/pmc_traffic.py --print $O/${TAG}_pmc_traffic.json ;;
    trace)   (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/proft && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/proft -- \
               python $R/bench.py ${arg:---steps 40 --warmup 5 --no-cpu-baseline --no-mip --no-kilo --no-unbounded --no-render --no-f16 --no-strict --no-extra} > /tmp/bt.log 2>&1; tail -c 300 /tmp/bt.log)
             python tools/trace_window.py $(ls /tmp/proft/*/*kernel_trace.csv | head -1) spans > $O/${TAG}_trace_spans.txt
             python tools/trace_window.py $(ls /tmp/proft/*/*kernel_trace.csv | head -1) ${TRACE_WIN:-0.55} | tee $O/${TAG}_trace_normal_iteration.txt ;;
    scatter) timeout 900 python tools/microbench_scatter3.py $arg 2>&1 | tee $O/${TAG}_microbench_scatter3.txt ;;
    fwd)     timeout 900 python tools/microbench_fwd3.py $arg 2>&1 | tee $O/${TAG}_microbench_fwd3.txt ;;
    py)      timeout 1200 python $arg 2>&1 | tee -a $O/${TAG}_py.txt ;;
    *)       echo "unknown stage $st" ;;
  esac
done
