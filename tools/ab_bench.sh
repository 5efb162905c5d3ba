# A/B of the training step on the GPU box: usage  bash tools/ab_bench.sh <tag> "ENV1=.. ENV2=.." ["ENV.." ...]
# prints value / ms_per_step per variant and stores the JSON lines under gpurun_out/<tag>_ab_<i>.json
TAG=${1:-ab}; shift
i=0
for envs in "" "$@"; do
  env $envs python bench.py --steps 400 --warmup 40 --no-cpu-baseline > gpurun_out/${TAG}_ab_$i.json 2> gpurun_out/${TAG}_ab_$i.err
  python - <<PY
import json
try:
    d = json.loads(open('gpurun_out/${TAG}_ab_$i.json').read().strip().splitlines()[-1])
    r = d.get('roofline', {})
    print('[%s] %-50s value %.0f  ms/step %.4f  dominant %.1f us  second %.1f us' % ('$TAG', '$envs' or 'default', d['value'], d['ms_per_step'],
          r.get('avg_launch_us', 0), r.get('second_kernel', {}).get('avg_launch_us', 0)))
except Exception as e:
    print('[%s] %s FAILED: %s' % ('$TAG', '$envs', e)); print(open('gpurun_out/${TAG}_ab_$i.err').read()[-800:])
PY
  i=$((i+1))
done
