# SQ counters of the row-panel kernel (one --pmc pass, --kernel-trace only): bash tools/pmc_panel.sh <tag>
TAG=${1:-panel}
OUT=$PWD/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_IFETCH SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_SMEM" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"; do
  n=$((n+1))
  rm -rf $OUT/fp_pmc_$n
  rocprofv3 --kernel-trace --pmc $set -d $OUT/fp_pmc_$n -o p -- python $OUT/../bench.py --steps 20 --warmup 5 --no-cpu-baseline --prewarm-s 0 > $OUT/${TAG}_pmc_$n.log 2>&1
  python $OUT/../tools/pmc_summary.py $OUT/fp_pmc_$n/p_results.db $OUT/${TAG}_pmc_$n.csv 0
  grep panel_t1 $OUT/${TAG}_pmc_$n.csv | cut -d, -f5- 
  rm -rf $OUT/fp_pmc_$n
done
