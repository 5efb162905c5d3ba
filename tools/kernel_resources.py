"""Register / LDS / occupancy table of every kernel of libpyprob_amd (the compiler's kernel-resource-usage remarks, the
same ones pyprob_amd/build.py guards the VGPR budgets with). No GPU needed:
    python tools/kernel_resources.py [--md] > profiles/<tag>_kernel_resources.txt"""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pyprob_amd import build as B  # noqa: E402

FIELDS = [('VGPRs', 'VGPRs'), ('AGPRs', 'AGPRs'), ('TotalSGPRs', 'SGPRs'), ('ScratchSize [bytes/lane]', 'scratch B/lane'),
          ('Occupancy [waves/SIMD]', 'waves/SIMD'), ('LDS Size [bytes/block]', 'LDS B/block'), ('VGPRs Spill', 'VGPR spill')]


def demangle(names):
    r = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True)
    return r.stdout.splitlines() if r.returncode == 0 else names


def remarks(src, tmp):
    cmd = [B._hipcc()] + B.FLAGS + ['-Rpass-analysis=kernel-resource-usage', '-c', os.path.join(B.CSRC, src), '-o',
                                    os.path.join(tmp, src + '.o')]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    out, cur = [], None
    for line in err.splitlines():
        m = re.search(r'remark: Function Name: (\S+)', line)
        if m:
            cur = dict(file=src, name=m.group(1))
            out.append(cur)
            continue
        m = re.search(r'remark:\s+([A-Za-z \[\]/]+): (\S+) \[-Rpass', line)
        if m and cur is not None:
            cur[m.group(1).strip()] = m.group(2)
    return out


def main():
    md = '--md' in sys.argv
    with tempfile.TemporaryDirectory() as tmp, ThreadPoolExecutor(max_workers=4) as ex:
        rows = [r for rs in ex.map(lambda s: remarks(s, tmp), B.SOURCES) for r in rs]
    names = demangle([r['name'] for r in rows])
    for r, n in zip(rows, names):
        n = n.replace('(anonymous namespace)::', '')      # (a '(' inside the name would otherwise cut it off below)
        n = re.sub(r'\(.*$', '', n).replace('void ', '').replace('pp::', '')
        r['short'] = n
    rows.sort(key=lambda r: (r['file'], r['short']))
    head = ['file', 'kernel'] + [h for _, h in FIELDS]
    table = [[r['file'], r['short']] + [r.get(k, '') for k, _ in FIELDS] for r in rows]
    if md:
        print('| ' + ' | '.join(head) + ' |')
        print('|' + '---|' * len(head))
        for t in table:
            print('| ' + ' | '.join(t) + ' |')
    else:
        w = [max(len(str(x)) for x in col) for col in zip(head, *table)]
        for t in [head] + table:
            print('  '.join(str(x).ljust(n) for x, n in zip(t, w)))


if __name__ == '__main__':
    main()
