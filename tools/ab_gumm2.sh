# A/B of the ragged (GUMM) training step: bash tools/ab_gumm2.sh <tag> "ENV=.." ...
TAG=${1:-abg}; shift
i=0
for envs in "" "$@"; do
  env $envs python bench.py --workload train_gumm --steps 150 --warmup 20 --no-cpu-baseline > gpurun_out/${TAG}_abg_$i.json 2> gpurun_out/${TAG}_abg_$i.err
  python - <<PY
import json
try:
    d = json.loads(open('gpurun_out/${TAG}_abg_$i.json').read().strip().splitlines()[-1])
    print('[%s] %-50s value %.0f  ms/step %.4f' % ('$TAG', '$envs' or 'default', d['value'], d['ms_per_step']))
except Exception as e:
    print('[%s] %s FAILED: %s' % ('$TAG', '$envs', e)); print(open('gpurun_out/${TAG}_abg_$i.err').read()[-600:])
PY
  i=$((i+1))
done
