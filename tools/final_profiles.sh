# Round profile set: kernel-trace stats of the three workloads + the two PMC passes of the training step.
# usage (GPU box): bash tools/final_profiles.sh <tag>     -> gpurun_out/<tag>_*
TAG=${1:-r01_e}
OUT=/root/repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/fp_* 
for wl in train train_gumm is; do
  steps=200; [ $wl = is ] && steps=40
  rocprofv3 --kernel-trace --stats -d $OUT/fp_$wl -o p -- python /root/repo/bench.py --workload $wl --steps $steps --warmup 20 --no-cpu-baseline > $OUT/${TAG}_${wl}_profiled_run.log 2>&1
  python /root/repo/tools/rocprof_summary.py $OUT/fp_$wl/p_results.db $OUT/${TAG}_${wl}_kernel_stats.csv 12
done
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $OUT/fp_pmc_$c -o p -- python /root/repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/pmc_$c.log 2>&1
  python /root/repo/tools/pmc_summary.py $OUT/fp_pmc_$c/p_results.db $OUT/${TAG}_train_pmc_$c.csv 6
done
cd /root/repo
python bench.py --steps 400 --warmup 30 > $OUT/${TAG}_train_bench_line.json
python bench.py --workload train_gumm --steps 200 --warmup 30 --no-cpu-baseline > $OUT/${TAG}_gumm_bench_line.json
python bench.py --workload is --steps 50 --warmup 5 > $OUT/${TAG}_is_bench_line.json
rm -rf $OUT/fp_*
tail -c 1500 $OUT/${TAG}_train_bench_line.json
