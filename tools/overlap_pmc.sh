#!/bin/bash
# SQ / TA counters of the lookup and the fused-MLP forward (each alone: --pmc serialises kernels) from tools/overlap_probe.py
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for c in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" "TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TCP_PENDING_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"; do
  rm -rf /tmp/pmco; timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pmco -- python $R/tools/overlap_probe.py > /tmp/pmco.log 2>&1
  f=$(ls /tmp/pmco/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -z "$f" ]; then echo "counters [$c]: no output ($(tail -c 200 /tmp/pmco.log | tr '\n' ' '))"; continue; fi
  python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'][:26]
    if 'hashgrid_fwd' not in k and 'mlp_fwd' not in k: continue
    acc[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
for k in acc:
    print(k, {c: '%.4g' % (v / cnt[(k, c)]) for c, v in acc[k].items()})
PY
done
