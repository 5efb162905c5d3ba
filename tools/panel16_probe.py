"""The 16-row panel kernel (PP_PANEL=2) against the tile path (PP_PANEL=0), one forward + backward, per-tensor errors:
    python tools/panel16_probe.py [B] [dist]        (runs itself twice in subprocesses; P16_H=1024: the wider network)"""
import os
import subprocess
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))


def child(B, dist, out, H=512):
    import torch
    from helpers import synthetic_gum_arrays
    from pyprob_amd.engine import ICEngine
    from pyprob_amd.packed import PackedBatch
    from pyprob_amd.spec import NetSpec
    spec = NetSpec({'obs0': {'dim': 32}, 'obs1': {'dim': 32}}, lstm_dim=H)
    spec.add_address('mu', dist)
    arr = synthetic_gum_arrays(B, seed=3 + B)
    if dist == 'Uniform':
        arr['prior'] = np.tile(np.array([[-4.0, 6.0]], np.float32), (B, 1))
        arr['values'] = np.clip(arr['values'], -3.9, 5.9).astype(np.float32)
        arr['values'][::97] = 7.5
    eng = ICEngine(spec, device='cuda:0', seed=5)
    pb = PackedBatch.from_ragged(arr['trace_len'], arr['addr_idx'], arr['values'], arr['prior'], arr['obs'], 1).to(eng.device)
    res = {}
    for rep in range(3):
        l, lp = eng.loss(pb, backward=True, keep_lp=True)
        torch.cuda.synchronize()
    res['loss'] = l.cpu().numpy()
    res['lp'] = lp.cpu().numpy()
    for k, v in eng.grad_dict().items():
        res['g/' + k] = np.asarray(v)
    np.savez(out, **res)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == '--child':
        child(int(sys.argv[2]), sys.argv[3], sys.argv[4], int(os.environ.get('P16_H', '512')))
        sys.exit(0)
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    dist = sys.argv[2] if len(sys.argv) > 2 else 'Normal'
    outs = {}
    for mode in ('2', '0'):
        f = '/tmp/p16_%s.npz' % mode
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', str(B), dist, f],
                           env=dict(os.environ, PP_PANEL=mode, PP_DETERMINISTIC='0'), timeout=600)
        if r.returncode != 0:
            print('mode', mode, 'failed rc', r.returncode)
            sys.exit(1)
        outs[mode] = dict(np.load(f))
    a, b = outs['2'], outs['0']
    if os.environ.get('P16_ORACLE'):      # both against the float64 oracle (which of the two moved?)
        from helpers import synthetic_gum_arrays
        from oracle import ic_oracle as O
        from pyprob_amd.spec import NetSpec
        import torch
        from pyprob_amd.engine import ICEngine
        H = int(os.environ.get('P16_H', '512'))
        arr = synthetic_gum_arrays(B, seed=3 + B)
        if dist == 'Uniform':
            arr['prior'] = np.tile(np.array([[-4.0, 6.0]], np.float32), (B, 1))
            arr['values'] = np.clip(arr['values'], -3.9, 5.9).astype(np.float32)
            arr['values'][::97] = 7.5
        spec = NetSpec({'obs0': {'dim': 32}, 'obs1': {'dim': 32}}, lstm_dim=H)
        spec.add_address('mu', dist)
        eng = ICEngine(spec, device='cuda:0', seed=5)
        P = {k: v.numpy().astype(np.float64) for k, v in eng.state_dict().items()}
        ref = O.loss_and_grads(O.Net(P, ['obs0', 'obs1'], K=10), arr, ['mu'], [dist])
        print('oracle loss', ref['loss'], 'panel16', a['loss'], 'tiles', b['loss'])
        for n in sorted(ref['grads']):
            r = ref['grads'][n]
            m = max(np.abs(r).max(), 1e-30)
            print('%-62s max|ref| %.2e  panel16 err %.2e  tiles err %.2e' % (n, m, np.abs(a['g/' + n] - r).max() / m, np.abs(b['g/' + n] - r).max() / m))
        sys.exit(0)
    print('H', os.environ.get('P16_H', '512'), 'B', B, dist, 'loss', a['loss'], b['loss'])
    for k in sorted(a):
        x, y = a[k].astype(np.float64), b[k].astype(np.float64)
        fin = np.isfinite(y)
        err = np.abs(x[fin] - y[fin]).max() / max(np.abs(y[fin]).max(), 1e-30)
        worst = 0.0
        if x.size >= 1024 and fin.all():      # per 1024-element chunk, relative to the chunk's own magnitude (tests/test_gpu_panel.py)
            n = (x.size // 1024) * 1024
            xa, ya = x.reshape(-1)[:n].reshape(-1, 1024), y.reshape(-1)[:n].reshape(-1, 1024)
            den = np.maximum(np.abs(ya).max(axis=1), 1e-6 * np.abs(y).max())
            worst = float((np.abs(xa - ya).max(axis=1) / den).max())
        print('%-70s max|ref| %.3e  rel err %.2e  worst chunk %.2e %s' % (k, np.abs(y[fin]).max(), err, worst, '' if err < 3e-5 and worst < 2e-3 else '  <-----'))
