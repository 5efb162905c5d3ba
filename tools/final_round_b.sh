# Second GPU call of a round's measurement set (the first: tools/final_round.sh <tag>): bash tools/final_round_b.sh <tag>
#   the ragged step (--workload train_gumm): kernel stats, launch sequence, PMC FETCH / WRITE -> profiles/r06_gumm_traffic.json, its
#   bench line, the dropin workload's line; then the SQ counter passes of the training step -> profiles/r06_mfma_busy.json
TAG=${1:-r07A}
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/fg_ks
rocprofv3 --kernel-trace --stats -d $OUT/fg_ks -o p -- python $REPO/bench.py --workload train_gumm --steps 60 --warmup 10 --no-cpu-baseline > $OUT/${TAG}_gumm_ks.log 2>&1
python $REPO/tools/rocprof_summary.py $OUT/fg_ks/p_results.db $OUT/${TAG}_train_gumm_kernel_stats.csv > /dev/null
python - <<P
import sys; sys.path.insert(0, '$REPO/tools')
import rocprof_summary as R
R.sequence('$OUT/fg_ks/p_results.db', '$OUT/${TAG}_ragged_step_sequence.csv')
P
rm -rf $OUT/fg_ks
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/fg_pmc_$c
  rocprofv3 --kernel-trace --pmc $c -d $OUT/fg_pmc_$c -o p -- python $REPO/bench.py --workload train_gumm --steps 20 --warmup 5 --no-cpu-baseline > $OUT/${TAG}_gumm_pmc_$c.log 2>&1
  python $REPO/tools/pmc_summary.py $OUT/fg_pmc_$c/p_results.db $OUT/${TAG}_train_gumm_pmc_$c.csv 0
  rm -rf $OUT/fg_pmc_$c
done
# kernel stats of the 25-step command for the launch count (same command as the PMC passes)
rm -rf $OUT/fg_ks2
rocprofv3 --kernel-trace --stats -d $OUT/fg_ks2 -o p -- python $REPO/bench.py --workload train_gumm --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
python $REPO/tools/rocprof_summary.py $OUT/fg_ks2/p_results.db $OUT/${TAG}_train_gumm_kernel_stats_25.csv > /dev/null
rm -rf $OUT/fg_ks2
cd $REPO
python tools/profile_gumm_json.py $TAG $OUT/${TAG}_train_gumm_pmc_FETCH_SIZE.csv $OUT/${TAG}_train_gumm_pmc_WRITE_SIZE.csv $OUT/${TAG}_train_gumm_kernel_stats_25.csv 25 $OUT/${TAG}_ragged_step_sequence.csv
cp profiles/r06_gumm_traffic.json $OUT/${TAG}_r06_gumm_traffic.json
python bench.py --workload train_gumm --steps 60 --warmup 10 > $OUT/${TAG}_gumm_bench_line.json 2> $OUT/${TAG}_gumm_bench.err
python bench.py --workload dropin --steps 50 > $OUT/${TAG}_dropin_bench_line.json 2> $OUT/${TAG}_dropin_bench.err
tail -c 400 $OUT/${TAG}_gumm_bench_line.json; echo
PASSES=3 bash tools/pmc_kernel.sh ${TAG}m panel16 --steps 100 --warmup 10 --no-is > $OUT/${TAG}_mfma.log 2>&1
python tools/mfma_busy_json.py ${TAG}m $OUT
cp profiles/r06_mfma_busy.json $OUT/${TAG}_r06_mfma_busy.json
python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_20_5_last.json 2> $OUT/${TAG}_bench_20_5_last.err
tail -c 300 $OUT/${TAG}_bench_20_5_last.json; echo
