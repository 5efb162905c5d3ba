"""Per-workgroup timeline of wgrad_t1_kernel (+ the reduction jobs behind it) for one ragged training step (GPU box):
   python tools/wg_trace_wgrad.py      # per problem of the launch (pp_debug_wgrad_plan lists them): when its workgroups start, how
                                       # long the K loop and the epilogue take, when they end; resident workgroups over time"""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'tests'))
import torch
import bench
from helpers import synthetic_gumm_arrays
from pyprob_amd import lib as L
from pyprob_amd.packed import PackedBatch

lib = L.load()
dev = torch.device('cuda:0')
eng = bench.make_engine(512, dev, seed=123)
_, addresses = synthetic_gumm_arrays(8, seed=0, max_iter=6)
eng.add_addresses([(a, 'Uniform', None) for a in addresses])
batches = []
for i in range(4):
    arr, _ = synthetic_gumm_arrays(1024, seed=100 + i, max_iter=6)
    ids = np.array([eng.spec.address_id[addresses[j]] for j in arr['addr_idx']])
    batches.append(PackedBatch.from_ragged(arr['trace_len'], ids, arr['values'], arr['prior'], arr['obs'], len(eng.spec.addresses)).to(dev))
for i in range(12):
    eng.train_step(batches[i % 4], 1e-3)
torch.cuda.synchronize()
cap = 8192
buf = torch.zeros(8 * cap, dtype=torch.int64, device=dev)
for rep in range(2):
    buf.zero_()
    lib.pp_debug_wgtrace(buf.data_ptr(), cap, 1)
    eng.train_step(batches[rep], 1e-3)
    torch.cuda.synchronize()
    lib.pp_debug_wgtrace(None, 0, 0)
    t = buf.cpu().numpy().reshape(cap, 8)
    live = t[:, 1] > 0
    idx = np.nonzero(live)[0]
    t = t[live]
    t0 = t[:, 0].min()
    us = lambda x: (x - t0) / 100.0
    print('rep %d: %d workgroups stamped, launch span %.2f us' % (rep, len(t), us(t[:, 1].max())))
    for q in sorted(set(t[:, 2].tolist())):
        m = t[:, 2] == q
        s, e = us(t[m, 0]), us(t[m, 1])
        line = 'problem %3d: %4d wgs (ids %4d..%4d) start %6.2f .. %6.2f  end %6.2f .. %6.2f  duration mean %6.2f max %6.2f' % (
            q, m.sum(), idx[m].min(), idx[m].max(), s.min(), s.max(), e.min(), e.max(), (e - s).mean(), (e - s).max())
        if q < 100:
            k = us(t[m, 4])
            line += '  K loop mean %6.2f  epilogue mean %5.2f' % ((k - s).mean(), (e - k).mean())
        print(line)
    # how many workgroups are running at each microsecond
    span = int(us(t[:, 1].max())) + 1
    run = np.zeros(span + 1)
    for a, b in zip(us(t[:, 0]), us(t[:, 1])):
        run[int(a):int(b) + 1] += 1
    print('resident workgroups every 5 us:', ' '.join('%d' % run[i] for i in range(0, span, 5)))
