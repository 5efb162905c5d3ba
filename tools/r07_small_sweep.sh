#!/bin/bash
# The small-network statement kernel (csrc/is_step_small.hip) against the chain of GEMM launches: H = 32 / 64 / 128, one and two layers
# usage: [SHAPES='96 1,192 2'] tools/r07_small_sweep.sh [out_dir] [modes] [n ...]
cd /root/repo
dir=${1:-gpurun_out/r07b}; modes=${2:-fused,fused_rows,chain}; shift 2 2>/dev/null
ns=${@:-64 2000 45000 200000 1000000}
mkdir -p $dir
out=$dir/small_statement_sweep.jsonl; : > $out
IFS=',' read -ra shapes <<< "${SHAPES:-32 1,32 2,64 1,64 2,128 1,128 2}"
for hd in "${shapes[@]}"; do
  set -- $hd
  H=$1 DEPTH=$2 MODES=$modes timeout 600 python tools/is_step_bench.py $ns >> $out 2>$dir/err_$1_$2.log
done
python - "$out" <<'P'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    print('H=%-4d depth=%d n=%-8d %-10s %8.4f ms  %7.1f M particles/s  executed %5.1f TFLOP/s  state %6.1f GB/s' % (
        d['H'], d['depth'], d['n'], d['mode'], d['ms_median'], d['particles_per_s'] / 1e6, d['tflops_executed'], d['state_GBps']))
P
