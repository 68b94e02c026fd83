#!/bin/bash
# PMC counters of k_composite_train on the bench's steady-state batch (separate rocprofv3 --pmc passes, no trace domains)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export MB_ONLY_CAPTURED=1
: > $O/r2l_pmc_composite_train.txt
for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
         "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCC_REQ_sum TCC_EA_WRREQ_sum TCC_EA_RDREQ_sum" \
         "GRBM_GUI_ACTIVE TCC_BUSY_avr TA_BUSY_avr"; do
  d=/tmp/pmc_ct; rm -rf $d
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $d -- python $R/tools/microbench_composite.py > /tmp/p.log 2>&1
  f=$(ls $d/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then python $R/tools/pmc_kernel.py $f k_composite_train | tee -a $O/r2l_pmc_composite_train.txt; else echo "no counters for: $c"; tail -3 /tmp/p.log; fi
done
