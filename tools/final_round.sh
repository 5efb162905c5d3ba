# The measurement set of a round on ONE fresh box session (bash tools/final_round.sh <tag>): everything lands under gpurun_out/<tag>_*;
# tools/collect_profiles.sh <tag> installs it under profiles/ here.
TAG=${1:-s5z}
OUT=$PWD/gpurun_out
bash tools/profile_round.sh $TAG > $OUT/${TAG}_round.log 2>&1          # fresh driver line first, rocprof + PMC of the step, lines
tail -3 $OUT/${TAG}_round.log
bash tools/pmc_is_fused.sh $TAG > $OUT/${TAG}_is_fused.log 2>&1        # SQ counters of the particle pass -> r06_is_fused_valu.json
tail -1 $OUT/${TAG}_is_fused.log | cut -c1-400
bash tools/profile_is_step.sh $TAG > $OUT/${TAG}_is_step.log 2>&1      # statement kernel: rocprof + PMC traffic -> r06_is_pmc_traffic.json
tail -2 $OUT/${TAG}_is_step.log | cut -c1-300
bash tools/quick_is_seq.sh $TAG > /dev/null 2>&1                        # kernels of one replayed posterior call
for d in r06_is_fused_valu r06_is_pmc_traffic; do cp profiles/$d.json $OUT/${TAG}_$d.json 2>/dev/null; done
H=1024 MODES=fused,fused_rows,chain python tools/is_step_bench.py 45000 200000 2> /dev/null | grep '^{' > $OUT/${TAG}_h1024_statement_bench.jsonl
python bench.py --steps 20 --warmup 5 --no-is --no-cpu-baseline --lstm-dim 1024 > $OUT/${TAG}_train_h1024_bench_line.json 2> /dev/null
python bench.py --workload train_gumm --steps 60 --warmup 10 --no-cpu-baseline > $OUT/${TAG}_gumm_bench_line.json 2> /dev/null
python bench.py --workload is --no-cpu-baseline > $OUT/${TAG}_is_bench_line.json 2> /dev/null
python tools/panel_timeline.py 2>&1 | tail -4 > $OUT/${TAG}_panel16_timeline.txt
python tools/is_call_profile.py 2>&1 | grep "wall per" > $OUT/${TAG}_is_call_wall.txt
# the driver's line LAST too (now quoting the committed-profile figures of this very tree)
python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_20_5_final.json 2> $OUT/${TAG}_bench_20_5_final.err
python -m pytest tests -m gpu -q > $OUT/${TAG}_gpu_tests.log 2>&1
grep -E "passed|failed" $OUT/${TAG}_gpu_tests.log | tail -2
