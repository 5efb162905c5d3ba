import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from helpers import synthetic_gum_arrays
from pyprob_amd.engine import ICEngine
from pyprob_amd.spec import NetSpec
from pyprob_amd.packed import PackedBatch
def eng_():
    spec = NetSpec({'obs0': {'dim': 32}, 'obs1': {'dim': 32}}, lstm_dim=64); spec.add_address('mu', 'Normal')
    return ICEngine(spec, seed=4)
sizes = (256, 300, 128, 200, 256, 512, 128)
def batches(e):
    out = []
    for k, n in enumerate(sizes):
        arr = synthetic_gum_arrays(n, seed=30 + k)
        out.append(PackedBatch.from_ragged(arr['trace_len'], np.zeros(n, np.int64), arr['values'], arr['prior'], arr['obs'], 1).to(e.device))
    return out
lrs = [1e-3 * (1 + 0.1 * k) for k in range(len(sizes))]
res = {}
for tag in ('step1', 'step2', 'res1', 'res2', 'res_chunk'):
    e = eng_(); bs = batches(e)
    if tag.startswith('step'):
        for pb, lr in zip(bs, lrs): e.train_step(pb, lr, weight_decay=1e-5)
    elif tag == 'res_chunk':
        e.train_resident(bs[:3], lrs[:3], weight_decay=1e-5); e.train_resident(bs[3:], lrs[3:], weight_decay=1e-5)
    else:
        e.train_resident(bs, lrs, weight_decay=1e-5)
    torch.cuda.synchronize()
    res[tag] = e.params.cpu().numpy().astype(np.float64)
    res[tag + '_t'] = {n: e.tensor(n).cpu().numpy().astype(np.float64) for n in e.spec.tensors}
    res[tag + '_steps'] = e.tensor_step.cpu().numpy().copy()
base = res['step1']
for k in ('step2', 'res1', 'res_chunk'):
    print(k, 'rel L2 vs step1: %.3g' % (np.linalg.norm(res[k] - base) / np.linalg.norm(base)))
rows = sorted(((np.abs(res['res1_t'][n] - res['step1_t'][n]).max(), n) for n in res['step1_t']), reverse=True)
for d, n in rows[:8]:
    print('   %-60s max diff %.3g  (max |p| %.3g)' % (n, d, np.abs(res['step1_t'][n]).max()))
d = np.abs(res['res1_t']['_layers_lstm.weight_ih_l0'] - res['step1_t']['_layers_lstm.weight_ih_l0'])
print('W_ih diff by column block: E[0:64] %.3g smp[64:68] %.3g prev[68:140] %.3g cur[140:212] %.3g; rows with diff: %d of %d' % (d[:, :64].max(), d[:, 64:68].max(), d[:, 68:140].max(), d[:, 140:].max(), int((d.max(1) > 1e-6).sum()), d.shape[0]))
print('rows (gate blocks of 64): i %.3g f %.3g g %.3g o %.3g' % tuple(d[64*k:64*(k+1)].max() for k in range(4)))
print('tensor_step step1', res['step1_steps'][:24].tolist())
print('tensor_step res1 ', res['res1_steps'][:24].tolist())
