#!/bin/bash
# The Marsaglia program's lock-step posterior call on small networks: fused small-network statement against the chain
cd /root/repo; mkdir -p gpurun_out/r07h; out=gpurun_out/r07h/gumm_small_ab.txt; : > $out
for hd in "64 1" "32 2" "128 2"; do
  set -- $hd
  for f in 1 0; do
    H=$1 DEPTH=$2 PP_IS_STEP_FUSED=$f timeout 600 python tools/gumm_call_bench.py 200000 12 2>&1 | grep -v amdgpu.ids | tail -1 >> $out
  done
done
cat $out
