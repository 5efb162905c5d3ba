"""profiles/r06_mfma_busy.json from the per-launch PMC averages of `tools/pmc_kernel.sh <tag> panel16 --steps 100 --warmup 10 --no-is`
(passes 1-3: gpurun_out/<tag>_pmc_{1,2,3}.csv), stamped with the hash of the kernel sources:
    python tools/mfma_busy_json.py <tag> [csv directory]"""
import csv
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_sha

tag = sys.argv[1]
src = sys.argv[2] if len(sys.argv) > 2 else 'gpurun_out'
NAMES = {'panel16_kernel': 'panel', 'wgrad_t1_kernel': 'wgrad_group', 'obs_embed_fwd_kernel': 'obs_embed_fwd', 'adam_kernel': 'adam'}
KEEP = ['SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_INSTS_MFMA', 'SQ_INSTS_VALU', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_VMEM_WR', 'SQ_WAVE_CYCLES',
        'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_VALU']
kernels = {}
for p in (1, 2, 3):
    with open(os.path.join(src, '%s_pmc_%d.csv' % (tag, p))) as f:
        for row in csv.DictReader(f):
            key = next((v for k, v in NAMES.items() if k in row['kernel']), None)
            if key is None or row['counter'] not in KEEP:
                continue
            k = kernels.setdefault(key, {})
            k[row['counter']] = float(row['avg_value'])
            if row['counter'] == 'SQ_VALU_MFMA_BUSY_CYCLES':
                k['avg_duration_us'] = round(float(row['avg_duration_ns']) / 1e3, 2)
for k in kernels.values():
    k.setdefault('avg_duration_us', 0.0)
    busy = k.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)
    k['mfma_busy_frac'] = round(busy / (k['avg_duration_us'] * 1e-6 * 2.4e9 * 1024), 4) if k['avg_duration_us'] else 0.0
    if k.get('SQ_WAVE_CYCLES'):
        k['wait_inst_over_wave_cycles'] = round(k.get('SQ_WAIT_INST_ANY', 0.0) / k['SQ_WAVE_CYCLES'], 3)
doc = dict(csrc_sha=csrc_sha(),
           source='tools/pmc_kernel.sh %s panel16 --steps 100 --warmup 10 --no-is (rocprofv3 --kernel-trace --pmc, separate passes; '
                  'per-launch averages in profiles/%s_pmc_{1,2,3}.csv)' % (tag, tag),
           definition="mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (launch duration x 2.4 GHz x 1024 SIMDs): the share of the chip's MFMA "
                      "pipes' time that is busy (a v_mfma_f32_16x16x4_f32 / 32x32x2 holds its pipe 32 / 64 cycles)",
           kernels=kernels)
with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'r06_mfma_busy.json'), 'w') as f:
    json.dump(doc, f, indent=1)
print({k: v['mfma_busy_frac'] for k, v in kernels.items()})
