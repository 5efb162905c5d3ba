# kernel sequence of ONE posterior_results call through the API: bash tools/profile_is.sh <tag>
TAG=${1:-prof}
OUT=$PWD/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/fp_is
rocprofv3 --kernel-trace --stats -d $OUT/fp_is -o p -- python $OUT/../tools/is_api_probe.py 30 > $OUT/${TAG}_is_profiled_run.log 2>&1
python $OUT/../tools/rocprof_summary.py $OUT/fp_is/p_results.db $OUT/${TAG}_is_step_sequence.csv sequence is_fused_kernel > /dev/null
rm -rf $OUT/fp_is
cat $OUT/${TAG}_is_step_sequence.csv | cut -c1-150
