"""Gradients of ragged minibatches of several sizes with PP_DH_PARTIALS on / off (GPU box): python tools/dh_diag.py"""
import os, subprocess, sys, tempfile
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUN = r'''
import sys, numpy as np, torch
sys.path.insert(0, %(repo)r); sys.path.insert(0, %(repo)r + '/tests')
from helpers import synthetic_gumm_arrays
from pyprob_amd.engine import ICEngine
from pyprob_amd.spec import NetSpec
from pyprob_amd.dataset import PackedTraceDataset
arrays, addresses = synthetic_gumm_arrays(6000, seed=12, max_iter=5)
table = [(a, 'Uniform', None) for a in addresses]
ds = PackedTraceDataset.from_columns(['obs0', 'obs1'], [1, 1], arrays['trace_len'], table, arrays['addr_idx'], arrays['values'], arrays['prior'], arrays['obs'])
spec = NetSpec({'obs0': {'dim': 32}, 'obs1': {'dim': 32}}, lstm_dim=64)
for a in addresses: spec.add_address(a, 'Uniform')
eng = ICEngine(spec, device='cuda:0', seed=3)
rng = np.random.default_rng(5)
out = {}
for k, n in enumerate((256, 300, 17, 1, 500, 64)):
    ids = rng.choice(6000, size=n, replace=False)
    pb = ds.device_batch(ids, eng.spec, eng.device)
    l = eng.loss(pb, backward=True)
    torch.cuda.synchronize()
    out['loss_%%d' %% n] = l.cpu().numpy()
    for name, g in eng.grad_dict().items():
        out['g_%%d_%%s' %% (n, name.replace('.', '__'))] = g
np.savez(sys.argv[1], **out)
'''
def run(tag, **env):
    f = os.path.join(tempfile.gettempdir(), 'dh_%s.npz' % tag)
    subprocess.run([sys.executable, '-c', RUN % dict(repo=REPO), f], check=True, env=dict(os.environ, **env), timeout=600)
    return dict(np.load(f))
a, b = run('on'), run('off', PP_DH_PARTIALS='0')
for n in (256, 300, 17, 1, 500, 64):
    print('batch %d: loss %.7f / %.7f' % (n, a['loss_%d' % n][0], b['loss_%d' % n][0]))
    rows = []
    for k in a:
        if k.startswith('g_%d_' % n):
            s = np.abs(b[k]).max()
            rows.append((np.abs(a[k] - b[k]).max() / max(s, 1e-12), k, s))
    rows.sort(reverse=True)
    for r in rows[:3]:
        print('     %-70s rel err %.3g (max |g| %.3g)' % (r[1], r[0], r[2]))
