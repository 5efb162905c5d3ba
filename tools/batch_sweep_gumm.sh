# The ragged program (config 3's network) on larger minibatches: bash tools/batch_sweep_gumm.sh <tag>
TAG=${1:-r02}
OUT=gpurun_out; mkdir -p $OUT
LOG=$OUT/${TAG}_batch_sweep_gumm.log; : > $LOG
for B in 1024 2048 4096 8192; do
  timeout 90 python bench.py --workload train_gumm --batch $B --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json, sys
try:
    d = json.loads(sys.stdin.read().strip().splitlines()[-1])
    w = d['roofline'].get('whole_step', {})
    print('gumm B=$B  %.4f ms/step  %.2f M traces/s  whole-step frac %.3f' % (d['ms_per_step'], d['value'] / 1e6, w.get('frac', float('nan'))))
except Exception as e:
    print('gumm B=$B FAILED', e)
" >> $LOG
done
cat $LOG
