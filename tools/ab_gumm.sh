# A/B of the ragged (GUMM) training step: bash tools/ab_gumm.sh <tag> "ENV=.." ...
TAG=${1:-abg}; shift
i=0
for envs in "" "$@"; do
  env $envs python bench.py --workload train_gumm --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/${TAG}_gumm_$i.json 2> gpurun_out/${TAG}_gumm_$i.err
  python - <<PY
import json
try:
    d = json.loads(open('gpurun_out/${TAG}_gumm_$i.json').read().strip().splitlines()[-1])
    print('[%s] %-50s value %.0f  ms/step %.4f  wgrad %.1f us' % ('$TAG', '$envs' or 'default', d['value'], d['ms_per_step'], d['roofline'].get('dominant_kernel', {}).get('avg_launch_us', 0)))
except Exception as e:
    print('[%s] %s FAILED: %s' % ('$TAG', '$envs', e)); print(open('gpurun_out/${TAG}_gumm_$i.err').read()[-800:])
PY
  i=$((i+1))
done
