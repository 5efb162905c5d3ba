# The N-row statement on the H = 1024 network under rocprofv3 (kernel trace + the two PMC passes): bash tools/profile_is_step_h1024.sh <tag> [n]
TAG=${1:-s5u}; N=${2:-200000}
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/fp_h1k
H=1024 MODES=fused,fused_rows rocprofv3 --kernel-trace --stats -d $OUT/fp_h1k -o p -- python $REPO/tools/is_step_bench.py $N > $OUT/${TAG}_h1024_is_step_bench.jsonl 2> $OUT/${TAG}_h1024_ks.err
python $REPO/tools/rocprof_summary.py $OUT/fp_h1k/p_results.db $OUT/${TAG}_h1024_is_step_kernel_stats.csv > /dev/null
rm -rf $OUT/fp_h1k
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/fp_h1kpmc_$c
  H=1024 MODES=fused rocprofv3 --kernel-trace --pmc $c -d $OUT/fp_h1kpmc_$c -o p -- python $REPO/tools/is_step_bench.py $N > $OUT/${TAG}_h1024_pmc_$c.log 2>&1
  python $REPO/tools/pmc_summary.py $OUT/fp_h1kpmc_$c/p_results.db $OUT/${TAG}_h1024_is_step_pmc_$c.csv 0
  rm -rf $OUT/fp_h1kpmc_$c
done
cd $REPO
cut -c1-160 $OUT/${TAG}_h1024_is_step_kernel_stats.csv | head -8
grep -i "is_step_fused\|is_prep" $OUT/${TAG}_h1024_is_step_pmc_FETCH_SIZE.csv $OUT/${TAG}_h1024_is_step_pmc_WRITE_SIZE.csv | cut -c1-220
