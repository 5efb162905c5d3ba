# SQ counters of is_fused_kernel (the pass over the particles of a GUM posterior call): bash tools/pmc_is_fused.sh <tag>
#   gpurun_out/<tag>_is_fused_pmc_{1,2}.csv, gpurun_out/<tag>_r05_is_fused_valu.json (-> profiles/r05_is_fused_valu.json, stamped with
#   the hash of csrc/: bench.py quotes it as `is.particle_kernels` when it matches the running tree)
TAG=${1:-s5a}
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
n=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS GRBM_GUI_ACTIVE"; do
  n=$((n+1))
  rm -rf $OUT/fp_isf_$n
  rocprofv3 --kernel-trace --pmc $set -d $OUT/fp_isf_$n -o p -- python $REPO/tools/is_call_profile.py 1000000 40 > $OUT/${TAG}_is_fused_pmc_$n.log 2>&1
  python $REPO/tools/pmc_summary.py $OUT/fp_isf_$n/p_results.db $OUT/${TAG}_is_fused_pmc_$n.csv 0
  rm -rf $OUT/fp_isf_$n
done
cd $REPO
python - <<P
import csv, json, sys
sys.path.insert(0, '$REPO')
from bench import csrc_sha
vals = {}
dur = None
for n in (1, 2):
    for r in csv.DictReader(open('$OUT/${TAG}_is_fused_pmc_%d.csv' % n)):
        if 'is_fused_kernel' in r['kernel'] and int(r['dispatches']) >= 20 and r['counter'] not in vals:
            vals[r['counter']] = float(r['avg_value'])
            dur = float(r['avg_duration_ns'])
doc = dict(csrc_sha=csrc_sha(), particles=1000000, avg_duration_us_profiled=None if dur is None else round(dur / 1e3, 2), counters=vals,
           source='rocprofv3 --kernel-trace --pmc (two passes, tools/pmc_is_fused.sh $TAG) of python tools/is_call_profile.py 1000000 40; '
                  'per-kernel averages in ${TAG}_is_fused_pmc_1.csv, ${TAG}_is_fused_pmc_2.csv')
if 'SQ_INSTS_VALU' in vals and 'GRBM_GUI_ACTIVE' in vals and vals['GRBM_GUI_ACTIVE'] > 0:
    # VALU issue slots: a wave64 VALU instruction occupies its SIMD-32 for 2 cycles (transcendentals 8); 256 CUs x 4 SIMDs
    cyc = vals['GRBM_GUI_ACTIVE']
    doc['valu_issue_fraction'] = round(vals['SQ_INSTS_VALU'] * 2.0 / (cyc * 1024.0), 4)
    doc['valu_issue_fraction_note'] = ('SQ_INSTS_VALU x 2 issue cycles / (GRBM_GUI_ACTIVE cycles x 1024 SIMDs): a LOWER bound of the VALU '
                                       'pipe occupancy (transcendental and fp64 instructions take 4x / longer); SQ_ACTIVE_INST_VALU / '
                                       'SQ_WAVE_CYCLES = the share of wave time with a VALU instruction in flight')
if 'SQ_ACTIVE_INST_VALU' in vals and vals.get('SQ_WAVE_CYCLES'):
    doc['valu_active_over_wave_cycles'] = round(vals['SQ_ACTIVE_INST_VALU'] / vals['SQ_WAVE_CYCLES'], 4)
json.dump(doc, open('$OUT/${TAG}_r05_is_fused_valu.json', 'w'), indent=1)
json.dump(doc, open('profiles/r05_is_fused_valu.json', 'w'), indent=1)
print(json.dumps(doc))
P
