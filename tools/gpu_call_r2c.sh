#!/bin/bash
# round 2, call C: forward gather v2 (plain 8-B gathers, two-list balanced XCD mapping), scatter phase timing, PMC
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 300 python tools/microbench_hash.py fwd 2>&1 | grep -v "amdgpu.ids" | tee $O/r2c_microbench_fwd.txt
timeout 60 tools/scatter_timing 2>&1 | tee $O/r2c_scatter_timing.txt
cd /tmp && export TMPDIR=/tmp
export XR_CHILD=fwd XR_HG_FWD_MODE=16
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCP_TOTAL_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
           "TCC_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_BUSY_sum TCC_TAG_STALL_sum"; do
  i=$((i+1)); d=/tmp/pmc_$i; rm -rf $d
  timeout 120 rocprofv3 --pmc $set --output-format csv -d $d -- python $R/tools/microbench_hash.py > /tmp/p_$i.log 2>&1
  f=$(ls $d/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then python $R/tools/pmc_kernel.py "$f" k_hashgrid_fwd; else echo "set $i failed: $(tail -c 300 /tmp/p_$i.log)"; fi
done 2>&1 | tee $O/r2c_pmc_fwd_mode16.txt
