"""Which tensors differ between two PP_DETERMINISTIC=1 processes? (diagnostic)"""
import os, subprocess, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'tests'))
from test_gpu_holes import REPEAT
from pyprob_amd.spec import NetSpec
outs = []
for k in range(2):
    f = '/tmp/det_%d.npz' % k
    subprocess.run([sys.executable, '-c', REPEAT % dict(repo=REPO), f], check=True, env=dict(os.environ, PP_DETERMINISTIC='1'))
    outs.append(dict(np.load(f)))
spec = NetSpec({'obs0': {'dim': 32}, 'obs1': {'dim': 32}}, lstm_dim=512)
spec.add_address('mu', 'Normal')
for key in ('gum_grads_0', 'gum_params'):
    a, b = outs[0][key], outs[1][key]
    print(key, 'equal' if np.array_equal(a, b) else 'DIFFER')
    for name, (off, shape) in spec.tensors.items():
        n = int(np.prod(shape))
        d = np.abs(a[off:off + n] - b[off:off + n]).max()
        if d > 0:
            print('   %-60s maxdiff %.3e  (scale %.3e)' % (name, d, np.abs(a[off:off + n]).max()))
print('loss', outs[0]['gum_loss_0'], outs[1]['gum_loss_0'])
