# HBM traffic counters of the training step (separate passes, --kernel-trace only): bash tools/pmc_train.sh <tag> [ENV=..]
TAG=${1:-pmc}; shift
OUT=$PWD/gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/fp_pmc_$c
  env "$@" rocprofv3 --kernel-trace --pmc $c -d $OUT/fp_pmc_$c -o p -- python $OUT/../bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/${TAG}_pmc_$c.log 2>&1
  python $OUT/../tools/pmc_summary.py $OUT/fp_pmc_$c/p_results.db $OUT/${TAG}_train_pmc_$c.csv 8
  rm -rf $OUT/fp_pmc_$c
done
