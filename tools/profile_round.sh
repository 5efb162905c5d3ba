# One GPU call's worth of evidence for the training step: bash tools/profile_round.sh <tag>
#   gpurun_out/<tag>_train_kernel_stats.csv, _step_sequence.csv   rocprofv3 --kernel-trace --stats of bench.py
#   gpurun_out/<tag>_train_pmc_{FETCH,WRITE}_SIZE.csv             separate --pmc passes
#   gpurun_out/<tag>_r06_kernel_avgs.json, _r06_pmc_traffic.json  what bench.py quotes (written to profiles/r05_*.json on the box; tools/collect_profiles.sh installs them here, stamped with the hash of csrc/)
#   gpurun_out/<tag>_bench_20_5.json, _bench_default.json         the driver's command and the default command
TAG=${1:-s5a}
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT
# the driver's command FIRST, in the fresh session (a bench line measured right after a profiled or test run of the same box
# session shows every memory-latency-bound kernel ~1.6x slower); it quotes no rocprof / PMC figures yet
python $REPO/bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_20_5_fresh.json 2> $OUT/${TAG}_bench_20_5_fresh.err
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/fp_ks
rocprofv3 --kernel-trace --stats -d $OUT/fp_ks -o p -- python $REPO/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-is > $OUT/${TAG}_ks.log 2>&1
python $REPO/tools/rocprof_summary.py $OUT/fp_ks/p_results.db $OUT/${TAG}_train_kernel_stats.csv > /dev/null
python - <<P
import sys; sys.path.insert(0, '$REPO/tools')
import rocprof_summary as R
R.sequence('$OUT/fp_ks/p_results.db', '$OUT/${TAG}_step_sequence.csv')
P
rm -rf $OUT/fp_ks
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/fp_pmc_$c
  rocprofv3 --kernel-trace --pmc $c -d $OUT/fp_pmc_$c -o p -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-is > $OUT/${TAG}_pmc_$c.log 2>&1
  python $REPO/tools/pmc_summary.py $OUT/fp_pmc_$c/p_results.db $OUT/${TAG}_train_pmc_$c.csv 0
  rm -rf $OUT/fp_pmc_$c
done
cd $REPO
python tools/profile_json.py $TAG $OUT/${TAG}_train_kernel_stats.csv $OUT/${TAG}_train_pmc_FETCH_SIZE.csv $OUT/${TAG}_train_pmc_WRITE_SIZE.csv
cp profiles/r06_kernel_avgs.json $OUT/${TAG}_r06_kernel_avgs.json; cp profiles/r06_pmc_traffic.json $OUT/${TAG}_r06_pmc_traffic.json
python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_20_5.json 2> $OUT/${TAG}_bench_20_5.err
python bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/${TAG}_bench_default.err
tail -c 600 $OUT/${TAG}_bench_20_5.json; echo; python -c "
import json
for f in ('20_5', 'default'):
    d = json.loads(open('$OUT/${TAG}_bench_' + f + '.json').read().strip().splitlines()[-1])
    r = d['roofline']
    print(f, d['value'], d['ms_per_step'], d.get('ms_per_step_median'), r['kernel'][:30], r['avg_launch_us'], r.get('rocprof_avg_us'), r['frac'], r.get('traffic'), r['whole_step']['frac'], d.get('is', {}).get('value'))
"
