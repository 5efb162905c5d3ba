#!/usr/bin/env python3
"""profiles/r06_gumm_traffic.json: HBM bytes of ONE ragged training step (all launches) from the two PMC passes of
`python bench.py --workload train_gumm --steps 20 --warmup 5 --no-cpu-baseline` (tools/profile_round6.sh), stamped with the hash
of the kernel sources; bench.py quotes it on a match.
    python tools/profile_gumm_json.py <tag> <pmc_FETCH_SIZE.csv> <pmc_WRITE_SIZE.csv> <kernel_stats.csv> <steps run under the profiler>"""
import csv
import json
import os
import sys


def main(tag, fetch_csv, write_csv, stats_csv, steps, sequence_csv=None):
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, repo)
    from bench import csrc_sha
    steps = int(steps)
    tot = {}
    per_kernel = {}
    for path, col, mult in ((fetch_csv, 'fetch_kb_raw', 2.0), (write_csv, 'write_kb', 1.0)):
        for r in csv.DictReader(open(path)):
            kb = float(r['avg_value']) * int(r['dispatches'])
            tot[col] = tot.get(col, 0.0) + kb
            k = r['kernel'].replace('void pp::', '').replace('pp::', '').replace('(anonymous namespace)::', '').split('(')[0].split('<')[0]
            per_kernel[k] = per_kernel.get(k, 0.0) + mult * kb * 1024 / steps
    launches = sum(int(r['calls']) for r in csv.DictReader(open(stats_csv)) if r['kernel'] != 'TOTAL')
    in_step = None
    if sequence_csv and os.path.exists(sequence_csv):      # the launches between two Adam launches: ONE step, in order
        in_step = sum(1 for _ in csv.DictReader(open(sequence_csv)))
    doc = dict(csrc_sha=csrc_sha(), steps_profiled=steps,
               traffic_bytes_per_step=int((2.0 * tot['fetch_kb_raw'] + tot['write_kb']) * 1024 / steps),
               fetch_kb_raw_per_step=round(tot['fetch_kb_raw'] / steps, 1), write_kb_per_step=round(tot['write_kb'] / steps, 1),
               launches_per_step=in_step if in_step else round(launches / steps, 2),
               bytes_per_step_by_kernel={k: int(v) for k, v in sorted(per_kernel.items(), key=lambda kv: -kv[1])[:12]},
               gfx950_fetch_correction='FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM section); KB = 1024 bytes',
               source='rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/profile_round6.sh %s) of `python bench.py '
                      '--workload train_gumm --steps 20 --warmup 5 --no-cpu-baseline`: every kernel dispatch of the process / steps run '
                      '(warm-up included: %d)' % (tag, steps))
    json.dump(doc, open(os.path.join(repo, 'profiles', 'r06_gumm_traffic.json'), 'w'), indent=1)
    print(json.dumps({k: doc[k] for k in ('traffic_bytes_per_step', 'launches_per_step')}))


if __name__ == '__main__':
    main(*sys.argv[1:7])
