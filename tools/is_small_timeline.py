"""Phase stamps (clock64, shader cycles) of the small-network statement kernel (csrc/is_step_small.hip): workgroup 0 and the
middle workgroup, first and last wave.   H=64 DEPTH=1 python tools/is_small_timeline.py [n] [shared 0|1]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyprob_amd.engine import ICEngine
from pyprob_amd.is_engine import ISRunner
from pyprob_amd.ops import ops
from pyprob_amd.spec import NetSpec

n = int(sys.argv[1]) if len(sys.argv) > 1 else 45000
shared = int(sys.argv[2]) if len(sys.argv) > 2 else 0
H = int(os.environ.get('H', '64'))
D = int(os.environ.get('DEPTH', '1'))
NAMES = ['start', 'rows+smp', 'stage old h'] + sum([['K loop %d' % l, 'barrier', 'cell %d' % l, 'barrier'] for l in range(D)], []) + \
        ['head 1', 'barrier', 'head 2', 'barrier', 'draw']
spec = NetSpec({'obs0': {'dim': 32}, 'obs1': {'dim': 32}}, lstm_dim=H, lstm_depth=D)
eng = ICEngine(spec, device='cuda:0', seed=0)
eng.add_addresses([('x', 'Uniform', None), ('y', 'Uniform', None)])
run = ISRunner(eng)
run.init([8.0, 9.0])
dev = eng.device
h = (0.3 * torch.randn(D, n, H, device=dev)).contiguous()
c = torch.randn(D, n, H, device=dev).contiguous()
pv = torch.rand(n, device=dev) * 2 - 1
prior = torch.tensor([[-1.0, 1.0]], device=dev)
run._ensure_ws(n)
buf = torch.zeros(128, dtype=torch.int64, device=dev)
for rep in range(3):
    if rep == 1:
        eng.lib.pp_debug_timeline(buf.data_ptr())
    buf.zero_()
    ops.is_step(eng.params, run.ws, eng.net_handle, 1, 0, n, run.e_obs, pv, prior, h, c, 1 if shared else n, None, 3, 0)
    torch.cuda.synchronize()
    if rep == 0:
        continue
    t = buf[:64].view(4, 16).tolist()
    print('--- H = %d, depth %d, n = %d, shared state = %d (cycles; 2400 cycles = 1 us)' % (H, D, n, shared))
    for row, tag in zip(t, ('wg 0 first wave', 'wg 0 last wave', 'wg mid first wave', 'wg mid last wave')):
        k_max = min(len(NAMES), 16)
        print('%-18s ' % tag + '  '.join('%s %d' % (NAMES[k], row[k] - row[k - 1]) for k in range(1, k_max) if row[k]) +
              '   | total %d' % (max(row[:k_max]) - row[0]))
eng.lib.pp_debug_timeline(None)
