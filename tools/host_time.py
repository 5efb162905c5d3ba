"""Host-side cost of one training step (launch path only): time N un-synchronised steps, then the drain."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from pyprob_amd.packed import ColumnarDataset
dev = torch.device('cuda:0')
eng = bench.make_engine(512, dev, 1)
obs, mu, prior = bench.synth_gum_dataset(1024 * 8, dev, 1)
ds = ColumnarDataset(obs, mu, prior, 1024); cache = {}
b = ds.batch(0, 0, 1, cache)
for _ in range(30): eng.train_step(b, 1e-3)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(50): eng.train_step(b, 1e-3)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('host enqueue %.1f us/step, drain after enqueue %.1f us/step-equivalent, total %.1f us/step' % ((t1 - t0) / 50 * 1e6, (t2 - t1) / 50 * 1e6, (t2 - t0) / 50 * 1e6))
