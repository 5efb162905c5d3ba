# PMC counters of one kernel of a bench.py workload (separate --pmc passes, --kernel-trace only):
#   [PASSES=3] bash tools/pmc_kernel.sh <tag> <kernel-name substring> <bench.py arguments...>   (PASSES: only the first counter sets)
TAG=$1; KERNEL=$2; shift 2
OUT=$PWD/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
n=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum TCC_ATOMIC_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"; do
  n=$((n+1))
  [ $n -gt ${PASSES:-5} ] && break
  rm -rf $OUT/fp_pmc_$n
  rocprofv3 --kernel-trace --pmc $set -d $OUT/fp_pmc_$n -o p -- python $OUT/../bench.py "$@" --no-cpu-baseline --prewarm-s 0 > $OUT/${TAG}_pmc_$n.log 2>&1
  python $OUT/../tools/pmc_summary.py $OUT/fp_pmc_$n/p_results.db $OUT/${TAG}_pmc_$n.csv 0
  grep "$KERNEL" $OUT/${TAG}_pmc_$n.csv | cut -d, -f2- | cut -c1-120
  rm -rf $OUT/fp_pmc_$n
done
