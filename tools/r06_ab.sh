#!/bin/bash
# A/B runs of round 6 (GPU box): wgrad_t1 tile order, first-statement fence. Benches first (fresh session), tests last.
out=gpurun_out/${1:-r06c}; mkdir -p $out
for x in 1 0 1 0; do
  PP_WGRAD_XMAP=$x python bench.py --steps 200 --warmup 20 --no-is --no-cpu-baseline 2>/dev/null | tail -1 > $out/train_xmap${x}_$RANDOM.json
done
for x in 1 0 1 0; do
  PP_WGRAD_XMAP=$x python bench.py --workload train_gumm --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > $out/gumm_xmap${x}_$RANDOM.json
done
for x in 1 0 1 0; do
  PP_IS_FIRST_SYNC=$x python bench.py --workload is --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $out/is_sync${x}_$RANDOM.json
done
python - <<PY
import json,glob
for f in sorted(glob.glob('$out/*.json')):
    try:
        d=json.loads(open(f).read())
    except Exception as e:
        print(f, 'ERR', e); continue
    r=d.get('roofline',{})
    print(f.split('/')[-1], d['ms_per_step'], d.get('value'), r.get('wgrad_us'), r.get('avg_launch_us'), (r.get('dominant_kernel') or {}).get('avg_launch_us'), d['config'].get('ms_per_call'))
PY
python -m pytest tests/test_gpu_binding_session.py tests/test_gpu_kernels.py tests/test_gpu_path.py tests/test_gpu_optim.py tests/test_gpu_panel.py -m gpu -x -q > $out/tests.log 2>&1; tail -4 $out/tests.log
