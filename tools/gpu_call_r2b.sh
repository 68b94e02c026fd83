#!/bin/bash
# round 2, call B: hardware counters of the forward gather (what bounds it?)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "^\s*(Name|Counter)?\s*:?\s*\b(SQ|TA|TCP|TD|TCC|GRBM)_[A-Za-z0-9_]+" | grep -oE "(SQ|TA|TCP|TD|TCC|GRBM)_[A-Za-z0-9_]+" | sort -u > $O/r2b_counters.txt
wc -l $O/r2b_counters.txt
export XR_CHILD=fwd XR_HG_FWD_MODE=${FWD_MODE:-0}
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_WAVE_CYCLES SQ_INSTS_LDS" \
           "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCP_TOTAL_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum" \
           "TCC_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_BUSY_sum TCC_TAG_STALL_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1)); d=/tmp/pmc_$i; rm -rf $d
  timeout 120 rocprofv3 --pmc $set --output-format csv -d $d -- python $R/tools/microbench_hash.py > /tmp/p_$i.log 2>&1
  f=$(ls $d/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then python $R/tools/pmc_kernel.py "$f" k_hashgrid_fwd; else echo "set $i failed: $(tail -c 300 /tmp/p_$i.log)"; fi
done 2>&1 | tee $O/r2b_pmc_fwd_mode${FWD_MODE:-0}.txt
