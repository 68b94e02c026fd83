#!/bin/bash
# SQ counters of the fused-MLP kernels (tools/microbench_mlp.py): wave cycles, waits, LDS / VALU / MFMA activity
cd /tmp && export TMPDIR=/tmp
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_LDS SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY"; do
  rm -rf /tmp/pmc; rocprofv3 --pmc $c --output-format csv -d /tmp/pmc -- python ${GRAFT_REPO_ROOT:-/root/repo}/tools/microbench_mlp.py > /dev/null 2>&1
  f=$(ls /tmp/pmc/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'][:24]
    if 'mlp_bwd' not in k and 'mlp_fwd' not in k: continue
    acc[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
for k in acc:
    print(k, {c: '%.4g' % (v / cnt[(k, c)]) for c, v in acc[k].items()})
PY
done
