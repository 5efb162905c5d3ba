# quick A/B of the training step on the GPU box: bench line, panel timeline, parity probe   (bash tools/quick_panel.sh <tag>)
TAG=${1:-q}
python bench.py --steps 20 --warmup 5 --no-is --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python -c "
import json
d=json.loads(open('gpurun_out/${TAG}_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernel'][:20], d['roofline']['avg_launch_us'])
"
python tools/panel_timeline.py 2>&1 | tail -4 > gpurun_out/${TAG}_panel16_timeline.txt; cat gpurun_out/${TAG}_panel16_timeline.txt
timeout 300 python tools/panel16_probe.py 1003 Normal 2>&1 | grep -c "<-----"
