#!/bin/bash
# SQ counters of the small-network statement kernel (csrc/is_step_small.hip): H=64 DEPTH=1, 10^6 particles (separate --pmc passes)
REPO=$PWD; OUT=$PWD/gpurun_out/r07r; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
n=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS SQ_IFETCH SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
  n=$((n+1))
  rm -rf $OUT/pmc_$n
  H=${H:-64} DEPTH=${DEPTH:-1} MODES=fused timeout 600 rocprofv3 --kernel-trace --pmc $set -d $OUT/pmc_$n -o p -- python $REPO/tools/is_step_bench.py 1000000 > $OUT/pmc_$n.log 2>&1
  python $REPO/tools/pmc_summary.py $OUT/pmc_$n/p_results.db $OUT/small_pmc_$n.csv 0
  grep "is_step_small_kernel" $OUT/small_pmc_$n.csv | cut -d, -f5- 
  rm -rf $OUT/pmc_$n
done
