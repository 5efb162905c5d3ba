# the GUM posterior call through the API: bench record with and without pp_is_first_statement   (bash tools/quick_is.sh <tag>)
TAG=${1:-q}
for f in 1 0; do
PP_IS_FIRST=$f python bench.py --workload is --no-cpu-baseline > gpurun_out/${TAG}_is_first$f.json 2> gpurun_out/${TAG}_is_first$f.err
tail -2 gpurun_out/${TAG}_is_first$f.err | cut -c1-300
python -c "
import json
d=json.loads(open('gpurun_out/${TAG}_is_first$f.json').read().strip().splitlines()[-1])
r=d.get('is', d)
print('PP_IS_FIRST=$f', d.get('value'), {k: r.get(k) for k in ('particles_per_sec','ms_per_call','plan_replays','noplan','device_chain')})
"
done
