# Round profile set: kernel-trace stats of the three workloads, the two PMC passes of the training step, and the bench
# lines (train at H=512 and H=1024, ragged GUMM training, IS posterior).
# usage (GPU box): bash tools/final_profiles.sh <tag>     -> gpurun_out/<tag>_*
TAG=${1:-r02_final}
ROOT=$PWD
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for wl in train train_gumm is; do
  steps=200; [ $wl = is ] && steps=40
  rm -rf $OUT/fp_$wl
  rocprofv3 --kernel-trace --stats -d $OUT/fp_$wl -o p -- python $ROOT/bench.py --workload $wl --steps $steps --warmup 20 --no-cpu-baseline > $OUT/${TAG}_${wl}_profiled_run.log 2>&1
  python $ROOT/tools/rocprof_summary.py $OUT/fp_$wl/p_results.db $OUT/${TAG}_${wl}_kernel_stats.csv 12
  rm -rf $OUT/fp_$wl
done
cd $ROOT
bash tools/pmc_train.sh $TAG > $OUT/${TAG}_pmc.log 2>&1
python bench.py --steps 400 --warmup 30 > $OUT/${TAG}_train_bench_line.json 2> $OUT/${TAG}_train_bench.err
python bench.py --lstm-dim 1024 --steps 200 --warmup 30 --no-cpu-baseline > $OUT/${TAG}_train_h1024_bench_line.json 2>> $OUT/${TAG}_train_bench.err
python bench.py --workload train_gumm --steps 200 --warmup 30 > $OUT/${TAG}_gumm_bench_line.json 2>> $OUT/${TAG}_train_bench.err
python bench.py --workload is --steps 50 --warmup 5 > $OUT/${TAG}_is_bench_line.json 2>> $OUT/${TAG}_train_bench.err
tail -c 2500 $OUT/${TAG}_train_bench_line.json
