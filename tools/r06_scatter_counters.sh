#!/bin/bash
# SQ / LDS / TCP counters of the table scatter (k_scatter_bin3, k_scatter_accum3, k_scatter_dense_rl, k_scatter_fold) and of the
# lookup inside the real training loop: which unit is saturated.  One --pmc pass per counter group (no trace domains beside it).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 48 --warmup 5 --no-cpu-baseline --no-mip --no-kilo --no-unbounded --no-render --no-f16 --no-strict --no-extra"
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
         "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
         "SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" \
         "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
         "SQ_WAVES GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_LDS_ATOMIC_RETURN" \
         "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  rm -rf /tmp/pmc; timeout 600 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc -- python $R/bench.py $ARGS > /tmp/pmc.log 2>&1
  f=$(ls /tmp/pmc/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -z "$f" ]; then echo "no counters for: $c"; tail -3 /tmp/pmc.log; continue; fi
  python - "$f" <<'PY'
import csv, sys, collections
want = ('k_scatter_bin3', 'k_scatter_accum3', 'k_scatter_dense_rl', 'k_scatter_fold', 'k_hashgrid_fwd', 'k_fwd3', 'k_nerf_mlp', 'k_composite')
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name']
    k = next((w for w in want if w in n), None)
    if k is None: continue
    acc[n[:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    print('%-40s' % k, {c: '%.4g (n=%d)' % (sum(v[-200:]) / len(v[-200:]), len(v)) for c, v in acc[k].items()})
PY
done
