"""Per-tensor comparison of the compact-row path with the full-width path (GPU box): python tools/compact_diag.py"""
import os, subprocess, sys, tempfile
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'tests'))
RUN = r'''
import sys, numpy as np, torch
sys.path.insert(0, %(repo)r); sys.path.insert(0, %(repo)r + '/tests')
from helpers import synthetic_gumm_arrays
from pyprob_amd.engine import ICEngine
from pyprob_amd.packed import PackedBatch
from pyprob_amd.spec import NetSpec
import json
H, n, depth = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
spec = NetSpec({'obs0': {'dim': 32}, 'obs1': {'dim': 32}}, lstm_dim=H, lstm_depth=depth)
arr, addresses = synthetic_gumm_arrays(n, seed=4, max_iter=4)
for a in addresses: spec.add_address(a, 'Uniform')
eng = ICEngine(spec, device='cuda:0', seed=5)
ids = np.array([spec.address_id[addresses[j]] for j in arr['addr_idx']])
pb = PackedBatch.from_ragged(arr['trace_len'], ids, arr['values'], arr['prior'], arr['obs'], len(spec.addresses)).to(eng.device)
l = eng.loss(pb, backward=True)
torch.cuda.synchronize()
g = eng.grad_dict()
np.savez(sys.argv[1], loss=l.cpu().numpy(), **{k.replace('.', '__'): v for k, v in g.items()})
'''
def run(tag, H, n, depth, **env):
    f = os.path.join(tempfile.gettempdir(), 'diag_%s.npz' % tag)
    subprocess.run([sys.executable, '-c', RUN % dict(repo=REPO), f, str(H), str(n), str(depth)], check=True,
                   env=dict(os.environ, **env), timeout=600)
    return dict(np.load(f))
for (H, n, depth) in ((256, 700, 1), (128, 900, 2)):
    a = run('c', H, n, depth)
    b = run('l', H, n, depth, PP_ADDR_BIAS='0')
    b2 = run('l2', H, n, depth, PP_ADDR_BIAS='0')
    d = run('d', H, n, depth, PP_DETERMINISTIC='1')
    gmax = max(np.abs(v).max() for k, v in b.items() if k != 'loss')
    print('H=%d n=%d depth=%d loss compact %.7f legacy %.7f det %.7f  global max |grad| %.4g' % (H, n, depth, a['loss'][0], b['loss'][0], d['loss'][0], gmax))
    rows = []
    for k in b:
        if k == 'loss': continue
        e = np.abs(a[k] - b[k]).max(); e2 = np.abs(b2[k] - b[k]).max(); e3 = np.abs(d[k] - b[k]).max()
        rows.append((e / gmax, k, e, np.abs(b[k]).max(), e2 / gmax, e3 / gmax))
    rows.sort(reverse=True)
    for r in rows[:12]:
        print('  %-70s err/gmax %.3g  abs %.3g  tensor max %.3g | legacy-vs-legacy %.3g  det-vs-legacy %.3g' % (r[1], r[0], r[2], r[3], r[4], r[5]))
