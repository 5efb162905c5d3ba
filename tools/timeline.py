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
for _ in range(20): eng.train_step(b, 1e-3)
buf = torch.zeros(128, dtype=torch.int64, device=dev)
lib.pp_debug_timeline(buf.data_ptr())
eng.train_step(b, 1e-3); torch.cuda.synchronize()
t = buf.tolist()
names = ['start', 'staged', 'a1 loaded', 'y done', 'mixture done', 'dy written', 'end']
for blk, off in ((0, 0), (100, 8)):
    print('workgroup', blk, ' '.join('%s +%d' % (names[k], t[off + k] - t[off + k - 1]) for k in range(1, 7)), ' total cycles', t[off + 6] - t[off])
