"""Per-workgroup timeline of the grouped async GEMM launches of one training step (GPU box):
   python tools/wg_trace.py [mode]     mode 1: weight-gradient group (+ reduction jobs), 2: data-gradient products (dX)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pyprob_amd import lib as L
from pyprob_amd.packed import ColumnarDataset

mode = int(sys.argv[1]) if len(sys.argv) > 1 else 1
lib = L.load()
dev = torch.device('cuda:0')
eng = bench.make_engine(512, dev, seed=123)
obs, mu, prior = bench.synth_gum_dataset(1024 * 16, dev, seed=1)
ds = ColumnarDataset(obs, mu, prior, 1024)
cache = {}
batches = [ds.batch(i, 0, 1, cache) for i in range(8)]
for i in range(20):
    eng.train_step(batches[i % 8], 1e-3)
torch.cuda.synchronize()
cap = 4096
buf = torch.zeros(8 * cap, dtype=torch.int64, device=dev)
for rep in range(3):
    buf.zero_()
    lib.pp_debug_wgtrace(buf.data_ptr(), cap, mode)
    eng.train_step(batches[rep], 1e-3)
    torch.cuda.synchronize()
    lib.pp_debug_wgtrace(None, 0, 0)
    t = buf.cpu().numpy().reshape(cap, 8)
    live = t[:, 1] > 0
    t = t[live]
    idx = np.nonzero(live)[0]
    t0 = t[:, 0].min()
    print('rep %d: %d workgroups stamped, launch span %.2f us' % (rep, len(t), (t[:, 1].max() - t0) / 100.0))
    for q in sorted(set(t[:, 2].tolist())):
        m = t[:, 2] == q
        s, e = (t[m, 0] - t0) / 100.0, (t[m, 1] - t0) / 100.0
        print('   problem %3d: %4d wgs (ids %4d..%4d)  start %.2f .. %.2f us   end %.2f .. %.2f us   mean duration %.2f us  max %.2f'
              % (q, m.sum(), idx[m].min(), idx[m].max(), s.min(), s.max(), e.min(), e.max(), (e - s).mean(), (e - s).max()))
        if q < 100 and (t[m, 4] > 0).all():
            ph = [(t[m, k] - t[m, 0]).mean() / 100.0 for k in (4, 5, 6, 7, 1)]
            print('        phases (mean, us after start): operands ready %.2f | loads issued %.2f | first slab landed %.2f | K loop done %.2f | end %.2f'
                  % tuple(ph))
