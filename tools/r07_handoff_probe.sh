#!/bin/bash
# Kernel-level A/B of the panel kernel's hand-off protocol (tools/micro/handoff_probe.hip) + its HBM-side traffic by PMC
REPO=$PWD; OUT=$PWD/gpurun_out/r07p; mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 tools/micro/handoff_probe.hip -o /tmp/handoff_probe 2> /dev/null || exit 1
timeout 300 /tmp/handoff_probe > $OUT/handoff_probe.txt 2>&1
cat $OUT/handoff_probe.txt
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/pmc_$c
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_$c -o p -- /tmp/handoff_probe > $OUT/pmc_$c.log 2>&1
  python $REPO/tools/pmc_summary.py $OUT/pmc_$c/p_results.db $OUT/handoff_probe_pmc_$c.csv 0
  rm -rf $OUT/pmc_$c
  cat $OUT/handoff_probe_pmc_$c.csv
done
