# kernels of one RAGGED training step (config 3) in launch order under rocprofv3   (bash tools/quick_gumm_seq.sh <tag>)
TAG=${1:-q}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/fp_kg -o p -- python $GRAFT_REPO_ROOT/bench.py --workload train_gumm --steps 100 --warmup 20 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py gpurun_out/fp_kg/p_results.db gpurun_out/${TAG}_train_gumm_kernel_stats.csv > /dev/null
python -c "
import sys; sys.path.insert(0,'tools')
import rocprof_summary as R
R.sequence('gpurun_out/fp_kg/p_results.db', 'gpurun_out/${TAG}_ragged_step_sequence.csv')
"
rm -rf gpurun_out/fp_kg
cat gpurun_out/${TAG}_ragged_step_sequence.csv | cut -c1-110
