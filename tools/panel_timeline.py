"""Phase stamps (clock64, shader cycles) of the row-panel kernel (csrc/panel.hip): workgroups 0 and 77, waves 0 and 5.
    python tools/panel_timeline.py            # config-2 step, B = 1024, H = 512"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from pyprob_amd import lib as L
from pyprob_amd.packed import ColumnarDataset

NAMES8 = ['start', 'staged', 'p1 input+cell', 'barrier', 'p2 units+W2 store', 'p2 barrier', 'p2 fixup', 'p3 tail product',
          'mixture', 'p4 dz1', 'barrier', 'p5 dh+cell bwd', 'p6 dX units', 'gsum+store']
# csrc/panel16.hip (PP_PANEL=2, the default): stamp k closes the interval named NAMES16[k]
NAMES16 = ['start', 'staged', 'p1 input+cell', 'barrier', 'p2 mfma', 'publish+tile16', 'fetch partners', 'z1+barrier', 'p3+barrier',
           'mixture+barrier', 'p4+barrier', 'p5 dh+cell bwd', 'p6 dX', 'reduce+exchange 2', 'embedding tail']
P16 = os.environ.get('PP_PANEL', '2') not in ('0', '1')
NAMES = NAMES16 if P16 else NAMES8
lib = L.load()
dev = torch.device('cuda:0')
eng = bench.make_engine(512, dev, seed=123)
obs, mu, prior = bench.synth_gum_dataset(1024 * 32, dev, seed=1000)
ds = ColumnarDataset(obs, mu, prior, 1024)
cache = {}
batches = [ds.batch(i, 0, 1, cache) for i in range(8)]
for i in range(20):
    eng.train_step(batches[i % 8], 1e-3)
buf = torch.zeros(128, dtype=torch.int64, device=dev)
lib.pp_debug_timeline(buf.data_ptr())
for rep in range(3):
    buf.zero_()
    eng.train_step(batches[rep], 1e-3)
    torch.cuda.synchronize()
    t = buf[:64].view(4, 16).tolist()
    print('--- step %d (cycles since the stamping wave started; 2400 cycles = 1 us)' % rep)
    for row, tag in zip(t, ('wg 0 wave 0', 'wg 0 wave 5', 'wg 77 wave 0', 'wg 77 wave 5')):
        base = row[0]
        if P16:
            print('%-13s ' % tag + '  '.join('%s %d' % (NAMES[k], row[k] - row[k - 1]) for k in range(1, 15)) + '   | total %d' % (row[14] - base))
        else:
            print('%-13s ' % tag + '  '.join('%s %d' % (NAMES[k], row[k] - row[k - 1]) for k in range(1, 14)) + '   | total %d | fixup: reduce+publish %d, wait+finish %d' % (row[13] - base, row[14] - row[5], row[6] - row[14]))

lib.pp_debug_timeline(None)
