# Round measurement set on the GPU box (GPU tests, kernel-trace stats of the training steps, the two PMC passes, bench lines
# at H = 512 / 1024, ragged, one-rank RCCL): bash tools/final_profiles.sh <tag>   -> gpurun_out/<tag>_*
TAG=${1:-r02_z}
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -6 > $OUT/${TAG}_gpu_tests.log; cat $OUT/${TAG}_gpu_tests.log
bash tools/profile_train.sh $TAG train > /dev/null 2>&1
bash tools/profile_train.sh $TAG train_gumm > /dev/null 2>&1
bash tools/pmc_train.sh $TAG > $OUT/${TAG}_pmc.log 2>&1
python bench.py --steps 400 --warmup 30 > $OUT/${TAG}_train_bench_line.json 2> $OUT/${TAG}_train_bench.err
python bench.py --lstm-dim 1024 --steps 200 --warmup 30 --no-cpu-baseline > $OUT/${TAG}_train_h1024_bench_line.json 2>> $OUT/${TAG}_train_bench.err
python bench.py --workload train_gumm --steps 200 --warmup 30 --no-cpu-baseline > $OUT/${TAG}_gumm_bench_line.json 2>> $OUT/${TAG}_train_bench.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 300 --warmup 30 --no-cpu-baseline > $OUT/${TAG}_train_rccl1_bench_line.json 2>> $OUT/${TAG}_train_bench.err
for f in train train_h1024 gumm train_rccl1; do python - <<PY
import json
try:
    d = json.loads(open('$OUT/${TAG}_${f}_bench_line.json').read().strip().splitlines()[-1])
    print('$f', d['value'], d['ms_per_step'], json.dumps(d.get('roofline', {}).get('whole_step', {}))[:160])
except Exception as e:
    print('$f FAILED', e)
PY
done
tail -3 $OUT/${TAG}_train_bench.err
