python bench.py --steps 20 --warmup 5 --no-is --no-cpu-baseline > gpurun_out/r05c_bench.json 2> gpurun_out/r05c_bench.err
python -c "
import json
d=json.loads(open('gpurun_out/r05c_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernel'][:20], d['roofline']['avg_launch_us'])
"
python tools/panel_timeline.py 2>&1 | tail -4 > gpurun_out/r05c_panel16_timeline.txt; cat gpurun_out/r05c_panel16_timeline.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/fp_ks -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-is > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py gpurun_out/fp_ks/p_results.db gpurun_out/r05c_train_kernel_stats.csv > /dev/null
python -c "
import sys; sys.path.insert(0,'tools')
import rocprof_summary as R
R.sequence('gpurun_out/fp_ks/p_results.db', 'gpurun_out/r05c_step_sequence.csv')
"
rm -rf gpurun_out/fp_ks
head -8 gpurun_out/r05c_train_kernel_stats.csv | cut -c1-200
