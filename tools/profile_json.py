#!/usr/bin/env python3
"""profiles/r06_kernel_avgs.json + profiles/r06_pmc_traffic.json (each stamped with `csrc_sha`, the hash of the kernel sources they
were measured on: bench.py quotes them only when it matches the running tree) (what bench.py quotes next to its live HIP-event timings) from
the per-kernel summaries of one `rocprofv3 --kernel-trace --stats` run and the two PMC passes (FETCH_SIZE, WRITE_SIZE) of
`python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-is`:
    python tools/profile_json.py <tag> <kernel_stats.csv> <pmc_FETCH_SIZE.csv> <pmc_WRITE_SIZE.csv>
"""
import csv
import json
import os
import sys

KEYS = (('panel16_kernel', 'panel'), ('panel_t1_kernel', 'panel'), ('wgrad_t1_kernel', 'wgrad_group'), ('gemm_f32_async_grouped_aux_kernel', 'wgrad_group'), ('obs_embed_fwd_kernel', 'obs_embed_fwd'),
        ('obs_embed_dgrad_kernel', 'obs_dgrad'), ('adam_kernel', 'adam'), ('gemm_f32_async_lstm_kernel', 'input_gemm'))
# algorithmic bytes per launch at B = 1024, H = 512, hid = 271, e = 64: what the launch must read and write once
ALG = {
    'panel': ('X 0.26 MB + weights once (W_ih[:, :64] and its transpose 1.0 MB, W1 and its transpose 1.1 MB, W2 0.03 MB) read; '
              'h 2.1 MB, dG 8.4 MB, a1 1.1 MB, dz1 1.1 MB, dy 0.1 MB, dX 0.26 MB written. Measured traffic above that: every L2 (one per XCD) '
              'fetches the 2.2 MB of weights (8 x, served by the Infinity Cache), and the two workgroups of a panel exchange their '
              'partial head-layer sums through 4.5 MB of write-through {value, tag} granules', 15.5e6),
    'wgrad_group': ('dG 8.4 MB, h 2.1 MB, a1 1.1 MB, dz1 1.1 MB, dy 0.1 MB, X 0.26 MB, observe-embedding activations 0.5 MB read; '
                    'W_ih table columns 1.2 MB read; gradients 2.4 MB written', 17.2e6),
}


def key_of(name):
    for sub, k in KEYS:
        if sub in name:
            return k
    return None


def main(tag, stats_csv, fetch_csv, write_csv):
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, repo)
    from bench import csrc_sha
    sha = csrc_sha()
    avgs, kern = {}, {}
    for r in csv.DictReader(open(stats_csv)):
        k = key_of(r['kernel'])
        if k and k not in avgs and int(r['calls']) >= 20:
            avgs[k] = float(r['avg_us'])
            kern[k] = '%s grid (%s,%s,%s)' % (r['kernel'].replace('(anonymous namespace)::', '').split('(')[0].replace('void pp::', ''),
                                              r['workgroups_x'], r['workgroups_y'], r['workgroups_z'])
    avgs['source'] = '%s (rocprofv3 --kernel-trace --stats of python bench.py)' % os.path.basename(stats_csv)
    avgs['csrc_sha'] = sha
    json.dump(avgs, open(os.path.join(repo, 'profiles', 'r06_kernel_avgs.json'), 'w'), indent=1)
    pm = {}
    for path, col in ((fetch_csv, 'FETCH_SIZE_KB_raw'), (write_csv, 'WRITE_SIZE_KB')):
        for r in csv.DictReader(open(path)):
            k = key_of(r['kernel'])
            if k and int(r['dispatches']) >= 20 and col not in pm.setdefault(k, {}):
                pm[k][col] = float(r['avg_value'])
    out = {}
    for k, d in pm.items():
        if len(d) < 2:
            continue
        d = dict(kernel=kern.get(k, k), **d)
        d['traffic_bytes_per_launch'] = int((2.0 * d['FETCH_SIZE_KB_raw'] + d['WRITE_SIZE_KB']) * 1024)
        d['algorithmic_bytes_per_launch'] = ALG[k][1] if k in ALG else None
        if k in ALG:
            d['algorithmic_bytes'] = ALG[k][0]
        out[k] = d
    doc = dict(csrc_sha=sha, source='rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/profile_round.sh %s) of '
                      '`python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-is`; per-kernel averages in %s, %s'
                      % (tag, os.path.basename(fetch_csv), os.path.basename(write_csv)),
               gfx950_fetch_correction='FETCH_SIZE reports 1/2 of wide coalesced reads on gfx950 (MI355X_MICROARCH.md, HBM section): '
                                       'doubled; counter unit KB = 1024 bytes', kernels=out)
    json.dump(doc, open(os.path.join(repo, 'profiles', 'r06_pmc_traffic.json'), 'w'), indent=1)
    print(json.dumps(avgs), '\n', json.dumps({k: v['traffic_bytes_per_launch'] for k, v in out.items()}))


if __name__ == '__main__':
    main(*sys.argv[1:5])
