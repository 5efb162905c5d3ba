#!/bin/bash
# Round 6, item 7 on one GPU: the two-part weight-gradient launch + early bucket on a side stream, on a ONE-rank RCCL group
out=gpurun_out/${1:-r06e}; mkdir -p $out
for x in 1 0 1 0; do
  PP_DP_OVERLAP=$x python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 400)) bench.py --gpus 1 --steps 200 --warmup 20 --no-is --no-cpu-baseline 2>$out/dp_overlap${x}.err | tail -1 > $out/dp_overlap${x}_$RANDOM.json
done
python - <<PY
import json,glob
for f in sorted(glob.glob('$out/dp_overlap*.json')):
    try:
        d=json.loads(open(f).read())
    except Exception as e:
        print(f, 'ERR', e); continue
    c=d['config']
    print(f.split('/')[-1], d['ms_per_step'], c.get('allreduce_us'), c.get('exposed_allreduce_us'), c.get('early_bucket_allreduce_us'), c.get('dp_overlap_ranges'), c.get('dp_exchange'))
PY
python -m pytest tests/test_gpu_dp_native.py tests/test_gpu_binding_session.py -m gpu -x -q > $out/tests.log 2>&1; tail -4 $out/tests.log
