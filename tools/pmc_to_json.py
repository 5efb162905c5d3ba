#!/usr/bin/env python3
"""profiles/<tag>_train_pmc_{FETCH,WRITE}_SIZE.csv (tools/pmc_summary.py of the two rocprofv3 --pmc passes) ->
profiles/<round>_pmc_traffic.json, the per-launch HBM traffic bench.py quotes in `roofline.traffic`.
FETCH_SIZE / WRITE_SIZE count KB; on gfx950 FETCH_SIZE reports half of wide coalesced reads (MI355X_MICROARCH.md, HBM
section) and is doubled."""
import csv
import json
import sys

# (key, kernel-name needle, algorithmic bytes per launch of the EXECUTED algorithm at B = 1024, H = 512, what)
KEYS = [('wgrad_group', 'gemm_f32_async_grouped_aux_kernel', 20200000,
         'last launch of the backward pass: weight-gradient tiles (dW_ih[:, :64] 2048x64x1024 reads dG 8.4 MB once, dW1 271x512x1024, '
         'dW2 30x271x1024, four observe-embedding leaves: operands 14.6 MB, outputs 1.2 MB) + reduction jobs (column sums 2 MB, '
         'table-column gradients: W_ih[:, 68:212] read 1.2 MB, dW_ih[:, 68:212] written 1.2 MB)'),
        ('input_gemm', 'gemm_f32_async_lstm_kernel', 9200000,
         'forward [E | s_prev] W_ih[:, :64]^T + bias + LSTM cell, 1024x2048x64: reads 0.8 MB, writes gates i, g, o 6.3 MB + h 2.1 MB'),
        ('dx_gemm', 'gemm_f32_async_grouped_kernel<false, true', 11000000,
         'dX[:, :64] = dG W_ih[:, :64], 1024x64x2048 in 16 stored K splits: reads dG (i, g, o columns) 6.3 MB + W 0.5 MB, writes 16 x 0.26 MB'),
        ('obs_embed_fwd', 'obs_embed_fwd_kernel', None, 'observe embedding + LSTM input rows + per-address bias vectors'),
        ('adam', 'adam_kernel', None, 'Adam pass over the flat buffers')]


def load(path):
    rows = {}
    for r in csv.DictReader(open(path)):
        rows.setdefault(r['kernel'], []).append(r)
    return rows


def main(tag, out):
    fetch, write = load('profiles/%s_train_pmc_FETCH_SIZE.csv' % tag), load('profiles/%s_train_pmc_WRITE_SIZE.csv' % tag)
    kernels = {}
    for key, needle, algo, what in KEYS:
        f = [r for k, v in fetch.items() if needle in k for r in v]
        w = [r for k, v in write.items() if needle in k for r in v]
        if not f or not w:
            continue
        f, w = max(f, key=lambda r: float(r['avg_duration_ns'])), max(w, key=lambda r: float(r['avg_duration_ns']))
        fk, wk = float(f['avg_value']), float(w['avg_value'])
        kernels[key] = dict(kernel='%s grid (%s,%s,%s): %s' % (needle, f['wg_x'], f['wg_y'], f['wg_z'], what), FETCH_SIZE_KB_raw=fk,
                            WRITE_SIZE_KB=wk, traffic_bytes_per_launch=int(round((2.0 * fk + wk) * 1024)),
                            algorithmic_bytes_per_launch=algo)
    doc = dict(source='rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/pmc_train.sh) of '
                      '`python bench.py --steps 20 --warmup 5 --no-cpu-baseline`; per-kernel averages in profiles/%s_train_pmc_*.csv' % tag,
               gfx950_fetch_correction='FETCH_SIZE reports 1/2 of wide coalesced reads on gfx950 (MI355X_MICROARCH.md, HBM section): '
                                       'doubled; counter unit KB = 1024 bytes', kernels=kernels)
    json.dump(doc, open(out, 'w'), indent=1)
    for k, v in kernels.items():
        print(k, v['traffic_bytes_per_launch'], v['algorithmic_bytes_per_launch'])


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
