# kernels of one training step in launch order under rocprofv3   (bash tools/quick_seq.sh <tag>)
TAG=${1:-q}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/fp_ks -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-is > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py gpurun_out/fp_ks/p_results.db gpurun_out/${TAG}_train_kernel_stats.csv > /dev/null
python -c "
import sys; sys.path.insert(0,'tools')
import rocprof_summary as R
R.sequence('gpurun_out/fp_ks/p_results.db', 'gpurun_out/${TAG}_step_sequence.csv')
"
rm -rf gpurun_out/fp_ks
head -5 gpurun_out/${TAG}_train_kernel_stats.csv | cut -d, -f1,2,7,8,9 | cut -c1-60,150-
