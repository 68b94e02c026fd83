#!/bin/bash
# prepared for the first GPU call of round 3 (not run yet): (1) rocprofv3 kernel stats of the bench on the final round-2 tree
# (the committed CSV predates the split forward), (2) hardware counters of the split forward -- what fills the two thirds of a
# tile's time that are neither matrix pipe nor conversion -- separate --pmc passes, no trace domains
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py --steps 128 --warmup 5 --no-cpu-baseline --no-mip --no-kilo --no-unbounded --no-f16 > /tmp/b.log 2>&1; tail -c 300 /tmp/b.log
cp $(ls /tmp/prof/*/*kernel_stats.csv | head -1) $O/r3a_bench_kernel_stats.csv; python $R/tools/kstats.py $O/r3a_bench_kernel_stats.csv | head -24
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1)); d=/tmp/pmc_$i; rm -rf $d
  timeout 120 rocprofv3 --pmc $set --output-format csv -d $d -- python $R/tools/microbench_mlp_fwd_split.py > /tmp/p_$i.log 2>&1
  f=$(ls $d/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then python $R/tools/pmc_kernel.py "$f" _Z17k_nerf_mlp_fwd_b3; python $R/tools/pmc_kernel.py "$f" void\ k_nerf_mlp_fwd_b3; else echo "set $i failed: $(tail -c 300 /tmp/p_$i.log)"; fi
done 2>&1 | tee $O/r3a_pmc_mlp_fwd_b3.txt
