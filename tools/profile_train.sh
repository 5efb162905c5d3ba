# kernel-trace stats of the training step: bash tools/profile_train.sh <tag> [workload]  -> gpurun_out/<tag>_<wl>_kernel_stats.csv
TAG=${1:-prof}; WL=${2:-train}
OUT=$PWD/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/fp_$WL
rocprofv3 --kernel-trace --stats -d $OUT/fp_$WL -o p -- python $OUT/../bench.py --workload $WL --steps 200 --warmup 20 --no-cpu-baseline > $OUT/${TAG}_${WL}_profiled_run.log 2>&1
python $OUT/../tools/rocprof_summary.py $OUT/fp_$WL/p_results.db $OUT/${TAG}_${WL}_kernel_stats.csv 16
if [ -n "$PP_SEQUENCE" ]; then python $OUT/../tools/rocprof_summary.py $OUT/fp_$WL/p_results.db $OUT/${TAG}_${WL}_step_sequence.csv sequence > /dev/null; fi
rm -rf $OUT/fp_$WL
cat $OUT/${TAG}_${WL}_kernel_stats.csv
