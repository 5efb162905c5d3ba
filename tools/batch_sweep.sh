# Throughput / whole-step roofline fraction against the minibatch size (config 2's network; the benchmark's B = 1024 is
# launch- and latency-bound): bash tools/batch_sweep.sh <tag>  -> gpurun_out/<tag>_batch_sweep.log
TAG=${1:-r02}
OUT=gpurun_out; mkdir -p $OUT
LOG=$OUT/${TAG}_batch_sweep.log; : > $LOG
for H in 512 1024; do
for B in 1024 2048 4096 8192 16384; do
  timeout 60 python bench.py --batch $B --lstm-dim $H --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json, sys
try:
    d = json.loads(sys.stdin.read().strip().splitlines()[-1])
    w = d['roofline'].get('whole_step', {})
    print('H=$H B=$B  %.4f ms/step  %.2f M traces/s  whole-step frac %.3f  dominant kernel frac %.3f' % (d['ms_per_step'], d['value'] / 1e6, w.get('frac', float('nan')), d['roofline']['frac']))
except Exception as e:
    print('H=$H B=$B FAILED', e)
" >> $LOG
done
done
cat $LOG
