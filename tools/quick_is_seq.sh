# kernels of one replayed GUM posterior call in launch order under rocprofv3   (bash tools/quick_is_seq.sh <tag>)
TAG=${1:-q}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/fp_is -o p -- python $GRAFT_REPO_ROOT/tools/is_call_profile.py 1000000 60 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python -c "
import sys; sys.path.insert(0,'tools')
import rocprof_summary as R
R.sequence('gpurun_out/fp_is/p_results.db', 'gpurun_out/${TAG}_is_call_sequence.csv', 'is_stats_combine_kernel', -40)
"
rm -rf gpurun_out/fp_is
