"""Effective shader clock right after (a) idle, (b) a burst of training steps, (c) a burst of big GEMMs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pyprob_amd import lib as L
from pyprob_amd.packed import ColumnarDataset

lib = L.load()
dev = torch.device('cuda:0')
out = torch.zeros(2, dtype=torch.int64, device=dev)
sink = torch.zeros(1, device=dev)


def probe(tag):
    lib.pp_debug_clock_probe(200000, out.data_ptr(), sink.data_ptr(), L.stream_ptr())
    torch.cuda.synchronize()
    c, w = out.tolist()
    print('%-34s shader cycles %9d  wall %8.1f us  -> %6.0f MHz' % (tag, c, w * 0.01, c / (w * 0.01)))


probe('cold')
probe('cold again')
eng = bench.make_engine(512, dev, 1)
obs, mu, prior = bench.synth_gum_dataset(1024 * 64, dev, 1)
ds = ColumnarDataset(obs, mu, prior, 1024)
cache = {}
batches = [ds.batch(i, 0, 1, cache) for i in range(64)]
for rep in range(3):
    for i in range(300):
        eng.train_step(batches[i % 64], 1e-3)
    probe('after 300 training steps')
A = torch.randn(8192, 8192, device=dev)
for rep in range(2):
    for _ in range(20):
        A @ A
    probe('after 20 torch 8192^3 matmuls')
time.sleep(1.0)
probe('after 1 s idle')
