# The N-row importance-sampling statement under rocprofv3: bash tools/profile_is_step.sh <tag> [n]
#   gpurun_out/<tag>_is_step_kernel_stats.csv           rocprofv3 --kernel-trace --stats of tools/is_step_bench.py (fused + chain)
#   gpurun_out/<tag>_is_step_pmc_{FETCH,WRITE}_SIZE.csv separate --pmc passes (fused kernel only)
#   profiles/r06_is_pmc_traffic.json                    HBM bytes per particle-statement, stamped with the hash of csrc/
TAG=${1:-s5a}; N=${2:-200000}
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/fp_isks
rocprofv3 --kernel-trace --stats -d $OUT/fp_isks -o p -- python $REPO/tools/is_step_bench.py $N > $OUT/${TAG}_is_step_bench.jsonl 2> $OUT/${TAG}_is_step_ks.err
python $REPO/tools/rocprof_summary.py $OUT/fp_isks/p_results.db $OUT/${TAG}_is_step_kernel_stats.csv > /dev/null
rm -rf $OUT/fp_isks
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/fp_ispmc_$c
  MODES=fused rocprofv3 --kernel-trace --pmc $c -d $OUT/fp_ispmc_$c -o p -- python $REPO/tools/is_step_bench.py $N > $OUT/${TAG}_is_step_pmc_$c.log 2>&1
  python $REPO/tools/pmc_summary.py $OUT/fp_ispmc_$c/p_results.db $OUT/${TAG}_is_step_pmc_$c.csv 0
  rm -rf $OUT/fp_ispmc_$c
done
cd $REPO
python - <<P
import csv, json, sys
sys.path.insert(0, '$REPO')
from bench import csrc_sha
n = $N
vals = {}
for c, col in (('FETCH_SIZE', 'fetch_kb_raw'), ('WRITE_SIZE', 'write_kb')):
    for r in csv.DictReader(open('$OUT/${TAG}_is_step_pmc_%s.csv' % c)):
        if 'is_step_fused_kernel' in r['kernel'] and int(r['dispatches']) >= 5:
            vals[col] = float(r['avg_value'])
            break
doc = dict(csrc_sha=csrc_sha(), source='rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of MODES=fused python tools/is_step_bench.py %d (H = 512)' % n,
           gfx950_fetch_correction='FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM section); KB = 1024 bytes', kernels={})
if len(vals) == 2:
    tr = (2.0 * vals['fetch_kb_raw'] + vals['write_kb']) * 1024
    doc['kernels']['is_step_fused'] = dict(particles=n, traffic_bytes_per_launch=int(tr), traffic_bytes_per_particle_statement=round(tr / n, 1),
                                           algorithmic_bytes_per_particle_statement=16 * 512 + 20,
                                           algorithmic_bytes='h and c read once and written once (4 x 2 KB), previous value, prior, value, log q (20 B); the 4.7 MB of weights once per launch',
                                           **vals)
json.dump(doc, open('profiles/r06_is_pmc_traffic.json', 'w'), indent=1)
json.dump(doc, open('$OUT/${TAG}_r06_is_pmc_traffic.json', 'w'), indent=1)
print(json.dumps(doc['kernels']))
P
cat $OUT/${TAG}_is_step_bench.jsonl
