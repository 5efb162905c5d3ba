"""Does a Python worker thread (prior generation) slow the native training call down? Times pp_train_steps runs alone,
with a numpy-busy thread, with a torch-busy thread (1 and default intra-op threads)."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
import torch
from helpers import synthetic_gum_arrays
from models import GaussianWithUnknownMean
from pyprob_amd.dataset import PackedTraceDataset
from pyprob_amd.engine import ICEngine
from pyprob_amd.spec import NetSpec
print('host cores', os.cpu_count(), 'torch threads', torch.get_num_threads(), 'affinity', len(os.sched_getaffinity(0)))
n, B = 262144, 1024
a = synthetic_gum_arrays(n, seed=1)
ds = PackedTraceDataset.from_columns(['obs0', 'obs1'], [1, 1], a['trace_len'], [('mu', 'Normal', None)], a['addr_idx'],
                                     a['values'], a['prior'], a['obs'])
spec = NetSpec({'obs0': {'dim': 32}, 'obs1': {'dim': 32}}, lstm_dim=512)
spec.add_address('mu', 'Normal')
eng = ICEngine(spec, seed=0)
rng = np.random.default_rng(0)
steps = [rng.choice(n, B, replace=False) for _ in range(64)]
model = GaussianWithUnknownMean()


def run(label, busy=None):
    stop = threading.Event()
    th = None
    if busy is not None:
        def loop():
            while not stop.is_set():
                busy()
        th = threading.Thread(target=loop, daemon=True)
        th.start()
    eng.train_run(ds, steps, [1e-3] * 64)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        l, s = eng.train_run(ds, steps, [1e-3] * 64)
        l.cpu()
    dt = (time.perf_counter() - t0) / 256
    stop.set()
    if th:
        th.join()
    print('%-40s %.1f us/step' % (label, dt * 1e6))


run('alone')
x = np.random.rand(1 << 17)
run('numpy sort thread', lambda: np.sort(x))
run('torch.normal thread (default threads)', lambda: torch.normal(torch.zeros(1 << 17), 1.0))
run('prior_traces_packed thread', lambda: model.prior_traces_packed(131072, ['obs0', 'obs1'], return_types=True))
torch.set_num_threads(1)
run('torch.normal thread (1 intra-op thread)', lambda: torch.normal(torch.zeros(1 << 17), 1.0))
run('prior_traces_packed thread, 1 intra-op', lambda: model.prior_traces_packed(131072, ['obs0', 'obs1'], return_types=True))
run('pure python thread', lambda: sum(range(20000)))
