"""Per-step cost of training from a packed in-memory dataset: the per-step Python loop (device_batch + loss + adam calls)
against the native run (pp_train_steps). python tools/train_loop_bench.py [feedforward]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
import torch
from helpers import synthetic_gum_arrays
from pyprob_amd.dataset import PackedTraceDataset
from pyprob_amd.engine import ICEngine
from pyprob_amd.spec import NetSpec

ff = len(sys.argv) > 1 and sys.argv[1] == 'feedforward'
n, B = 262144, 1024
a = synthetic_gum_arrays(n, seed=1)
ds = PackedTraceDataset.from_columns(['obs0', 'obs1'], [1, 1], a['trace_len'], [('mu', 'Normal', None)], a['addr_idx'],
                                     a['values'], a['prior'], a['obs'])
spec = NetSpec({'obs0': {'dim': 32}, 'obs1': {'dim': 32}}, lstm_dim=512, network='feedforward' if ff else 'lstm')
spec.add_address('mu', 'Normal')
eng = ICEngine(spec, seed=0)
rng = np.random.default_rng(0)
steps = [rng.choice(n, B, replace=False) for _ in range(256)]


def python_loop():
    for ids in steps:
        pb = ds.device_batch(ids, spec, eng.device)
        eng.loss(pb, backward=True)
        eng.adam_step(1e-3, zero_grads=True)


def native(chunk):
    for s in range(0, len(steps), chunk):
        l, st = eng.train_run(ds, steps[s:s + chunk], [1e-3] * len(steps[s:s + chunk]))
        l.cpu()


for name, fn in (('python per-step loop', python_loop), ('native run, 64 steps per call', lambda: native(64)),
                 ('native run, 256 steps per call', lambda: native(256)), ('native run, 1 step per call', lambda: native(1))):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / len(steps)
    print('%-32s %.1f us/step = %.2f M traces/s' % (name, dt * 1e6, B / dt / 1e6))
# host-only cost of a native run: time until the call returns (GPU still busy) for one 64-step run
torch.cuda.synchronize()
t0 = time.perf_counter()
eng.train_run(ds, steps[:64], [1e-3] * 64)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('one 64-step run: call returns after %.2f ms, GPU done after %.2f ms' % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
