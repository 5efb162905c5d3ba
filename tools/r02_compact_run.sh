# GPU-box run for the compact-row step: kernarg probe, the switch-by-switch parity tests, the whole GPU suite, A/B bench
# lines and a kernel-trace profile.   usage: bash tools/r02_compact_run.sh <tag>
TAG=${1:-r02_n}
OUT=gpurun_out
mkdir -p $OUT
( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -w tools/micro/kernarg_probe.hip -o /tmp/kernarg_probe && /tmp/kernarg_probe ) > $OUT/${TAG}_kernarg_probe.log 2>&1
cat $OUT/${TAG}_kernarg_probe.log
timeout 900 python -m pytest tests/test_gpu_compact.py -x -q 2>&1 | tail -25 > $OUT/${TAG}_compact_tests.log
cat $OUT/${TAG}_compact_tests.log
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -25 > $OUT/${TAG}_gpu_tests.log
cat $OUT/${TAG}_gpu_tests.log
bash tools/ab_bench.sh $TAG "PP_ADDR_BIAS=0" "PP_FUSE_CELL=0" "PP_FUSE_CELL_BWD=0" "PP_AUX_COLSUM=0" "PP_AUX_FUSED=0"
python bench.py --workload train_gumm --steps 100 --warmup 20 --no-cpu-baseline > $OUT/${TAG}_gumm_bench_line.json 2> $OUT/${TAG}_gumm_bench.err
cut -c1-330 $OUT/${TAG}_gumm_bench_line.json
bash tools/profile_train.sh $TAG train > /dev/null 2>&1
head -16 $OUT/${TAG}_train_kernel_stats.csv | cut -c1-60,200-400 
bash tools/profile_train.sh $TAG train_gumm > /dev/null 2>&1
