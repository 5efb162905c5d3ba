"""Probe: the ragged (GaussianUnknownMeanMarsaglia) training step on ONE fixed minibatch, launched eagerly and replayed as a
captured HIP graph - what the launch gaps of its ~25 dependent launches cost.  python tools/ragged_graph_probe.py [steps]"""
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
from helpers import synthetic_gumm_arrays          # noqa: E402
from pyprob_amd.engine import ICEngine             # noqa: E402
from pyprob_amd.packed import PackedBatch          # noqa: E402
from pyprob_amd.spec import NetSpec                # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 200
spec = NetSpec({'obs0': {'dim': 32}, 'obs1': {'dim': 32}}, lstm_dim=512)
_, addresses = synthetic_gumm_arrays(8, seed=0, max_iter=6)
for a in addresses:
    spec.add_address(a, 'Uniform')
eng = ICEngine(spec, device='cuda:0', seed=1)
arr, _ = synthetic_gumm_arrays(1024, seed=100, max_iter=6)
ids = np.array([eng.spec.address_id[addresses[j]] for j in arr['addr_idx']])
pb = PackedBatch.from_ragged(arr['trace_len'], ids, arr['values'], arr['prior'], arr['obs'], len(eng.spec.addresses)).to(eng.device)
out = {}
for _ in range(30):
    eng.train_step(pb, 1e-3)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    eng.train_step(pb, 1e-3)
torch.cuda.synchronize()
out['eager_ms'] = (time.perf_counter() - t0) / K * 1e3
eng.capture_train_step(pb, 1e-3)
for _ in range(30):
    eng.replay_train_step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    eng.replay_train_step()
torch.cuda.synchronize()
out['graph_ms'] = (time.perf_counter() - t0) / K * 1e3
print(json.dumps(out))
