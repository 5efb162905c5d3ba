#!/bin/bash
# In-kernel A/B of the panel kernel's hand-off protocol: PP_PANEL_HANDOFF=flag against the default granules
#   alternating bench runs (driver's command), rocprof kernel averages, PMC FETCH / WRITE of both
REPO=$PWD; OUT=$PWD/gpurun_out/r07w; mkdir -p $OUT
for rep in 1 2 3; do
  for m in granule flag; do
    PP_PANEL_HANDOFF=$m python bench.py --steps 20 --warmup 5 --no-is --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=l['roofline']
print('$m rep $rep: %.1f traces/s  %.4f ms/step  median %.4f  panel16 avg %.2f us' % (l['value'], l['ms_per_step'], l['config'].get('ms_per_step_median', 0), r['avg_launch_us']))" >> $OUT/ab.txt
  done
done
cd /tmp && export TMPDIR=/tmp
for m in granule flag; do
  rm -rf $OUT/ks_$m
  PP_PANEL_HANDOFF=$m rocprofv3 --kernel-trace --stats -d $OUT/ks_$m -o p -- python $REPO/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-is > $OUT/ks_$m.log 2>&1
  python $REPO/tools/rocprof_summary.py $OUT/ks_$m/p_results.db $OUT/kernel_stats_$m.csv > /dev/null
  rm -rf $OUT/ks_$m
  echo "== $m" >> $OUT/ab.txt; grep -E "panel16|wgrad_t1|adam_kernel|obs_embed_fwd" $OUT/kernel_stats_$m.csv | cut -c1-200 >> $OUT/ab.txt
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $OUT/pmc_${m}_$c
    PP_PANEL_HANDOFF=$m rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_${m}_$c -o p -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-is > $OUT/pmc_${m}_$c.log 2>&1
    python $REPO/tools/pmc_summary.py $OUT/pmc_${m}_$c/p_results.db $OUT/pmc_${m}_$c.csv 0
    rm -rf $OUT/pmc_${m}_$c
    grep panel16 $OUT/pmc_${m}_$c.csv | cut -d, -f1,5- | cut -c1-200 >> $OUT/ab.txt
  done
done
cat $OUT/ab.txt
