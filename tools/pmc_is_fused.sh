# SQ counters of is_fused_kernel (the pass over the particles of a GUM posterior call): bash tools/pmc_is_fused.sh <tag>
#   gpurun_out/<tag>_is_fused_pmc_{1,2}.csv, gpurun_out/<tag>_r06_is_fused_valu.json (-> profiles/r06_is_fused_valu.json, stamped with
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
# SIMD cycles of the launch: its average duration x the 2.4 GHz maximum clock x 256 CUs x 4 SIMDs - the clock under the
# profiler is lower (MI355X guide: 1.9-2.0 GHz), so both fractions are LOWER bounds. (GRBM_GUI_ACTIVE is collected but not used: it
# reads ~12x the launch's own cycles here - summed over XCDs / shader engines.)
if dur and 'SQ_INSTS_VALU' in vals:
    simd_cycles = dur * 1e-9 * 2.4e9 * 1024.0
    doc['simd_cycles_at_max_clock'] = round(simd_cycles)
    doc['valu_issue_fraction'] = round(vals['SQ_INSTS_VALU'] * 2.0 / simd_cycles, 4)
    doc['valu_issue_fraction_note'] = ('SQ_INSTS_VALU x 2 issue cycles (a wave64 instruction on a SIMD-32; transcendentals take 8) / SIMD '
                                       'cycles of the launch at the 2.4 GHz maximum clock: a lower bound')
    if 'SQ_ACTIVE_INST_VALU' in vals:
        doc['valu_busy_fraction'] = round(vals['SQ_ACTIVE_INST_VALU'] * 4.0 / simd_cycles, 4)
        doc['valu_busy_fraction_note'] = ('SQ_ACTIVE_INST_VALU (quad-cycles a VALU instruction is executing, summed over the waves) x 4 / '
                                          'SIMD cycles of the launch: the share of the VALU pipes\' time that is busy - the roofline '
                                          'fraction of this VALU-bound kernel')
if 'SQ_ACTIVE_INST_VALU' in vals and vals.get('SQ_WAVE_CYCLES'):
    doc['valu_active_over_wave_cycles'] = round(vals['SQ_ACTIVE_INST_VALU'] / vals['SQ_WAVE_CYCLES'], 4)
if vals.get('SQ_WAIT_ANY') and vals.get('SQ_WAVE_CYCLES'):
    doc['wait_any_over_wave_cycles'] = round(vals['SQ_WAIT_ANY'] / vals['SQ_WAVE_CYCLES'], 4)
json.dump(doc, open('$OUT/${TAG}_r06_is_fused_valu.json', 'w'), indent=1)
json.dump(doc, open('profiles/r06_is_fused_valu.json', 'w'), indent=1)
print(json.dumps(doc))
P
