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
names = ['start', 'staged', 'trace0 loaded', 'gF1', 'dz1', 'gF0+dzc', 'obs layers', 'all traces', 'wave turns', 'flush']
def show(tag):
    t = buf.tolist()
    print('%-28s' % tag, ' '.join('%s +%d' % (names[k], t[k] - t[k - 1]) for k in range(1, 10)), ' total', t[9] - t[0])
for _ in range(20): eng.train_step(b, 1e-3)
lib.pp_debug_timeline(buf.data_ptr())
for rep in range(3):
    eng.train_step(b, 1e-3); torch.cuda.synchronize(); show('train_step (after Adam)')
for rep in range(3):
    eng.loss(b, backward=True); torch.cuda.synchronize(); show('loss+bwd only (no Adam)')
