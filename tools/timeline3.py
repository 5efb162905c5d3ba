import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from pyprob_amd import lib as L
from pyprob_amd.packed import ColumnarDataset
lib = L.load(); dev = torch.device('cuda:0')
eng = bench.make_engine(512, dev, 1)
obs, mu, prior = bench.synth_gum_dataset(1024 * 8, dev, 1)
ds = ColumnarDataset(obs, mu, prior, 1024); cache = {}
b = ds.batch(0, 0, 1, cache)
buf = torch.zeros(128, dtype=torch.int64, device=dev)
for _ in range(20): eng.train_step(b, 1e-3)
lib.pp_debug_timeline(buf.data_ptr())
for rep in range(3):
    eng.loss(b); torch.cuda.synchronize()     # forward only: only the input GEMM stamps slots 0..4
    t = buf.tolist()[10:]
    print('input GEMM tile(1,1): init+loads issued +%d  first slab in LDS +%d  7 slabs +%d  epilogue +%d  total %d' % (t[1]-t[0], t[2]-t[1], t[3]-t[2], t[4]-t[3], t[4]-t[0]))
