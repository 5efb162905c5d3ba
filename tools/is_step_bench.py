"""One N-row importance-sampling statement (pp_is_step, per-particle LSTM state; BASELINE.json configs[3] network: H = 512):
the fused statement kernel (csrc/is_step_fused.hip) against the unfused chain (PP_IS_STEP_FUSED=0), HIP events on torch's
stream. usage: [H=512] [DEPTH=1] python tools/is_step_bench.py [n ...]   -> one JSON line per (n, mode)
(H = 32 / 64 / 128 with DEPTH 1..4: the small-network statement kernel, csrc/is_step_small.hip)"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pyprob_amd.engine import ICEngine
from pyprob_amd.is_engine import ISRunner
from pyprob_amd.ops import ops
from pyprob_amd.spec import NetSpec

H = int(os.environ.get('H', '512'))
DEPTH = int(os.environ.get('DEPTH', '1'))
ns = [int(a) for a in sys.argv[1:]] or [45000, 200000]
spec = NetSpec({'obs0': {'dim': 32}, 'obs1': {'dim': 32}}, lstm_dim=H, lstm_depth=DEPTH)
eng = ICEngine(spec, device='cuda:0', seed=0)
eng.add_addresses([('x', 'Uniform', None), ('y', 'Uniform', None)])
run = ISRunner(eng)
run.init([8.0, 9.0])
dev = eng.device
ad = eng.net.addrs[1]
for n in ns:
    h = (0.3 * torch.randn(DEPTH, n, H, device=dev)).contiguous()
    c = torch.randn(DEPTH, n, H, device=dev).contiguous()
    pv = torch.rand(n, device=dev) * 2 - 1
    prior = torch.tensor([[-1.0, 1.0]], device=dev)
    rows = torch.arange(n, device=dev, dtype=torch.int64)
    run._ensure_ws(n)
    flops_alg = n * (2.0 * (spec.lstm_in + H) * 4 * H + (DEPTH - 1) * 2.0 * 2 * H * 4 * H +
                     2.0 * (H * ad.hid + ad.hid * ad.n_out))
    if H >= 256:
        flops_exe = n * (2.0 * (8 + H) * 4 * H + 2.0 * (H * 288 + 272 * 32))                             # what the fused kernel multiplies
    else:      # is_step_small.hip: one 8-k slab of sample embedding + H per layer 0, 2 H per later layer; heads padded to 32 / 8
        flops_exe = n * (2.0 * (8 + H) * 4 * H + (DEPTH - 1) * 2.0 * 2 * H * 4 * H +
                         2.0 * (H * 32 * ((ad.hid + 31) // 32) + 8 * ((ad.hid + 7) // 8) * 32))
    for mode in os.environ.get('MODES', 'fused,fused_rows,chain').split(','):
        os.environ['PP_IS_STEP_FUSED'] = {'chain': '0', 'split': '3'}.get(mode, '2')      # (split: the two-launch statement)
        def call():
            if mode == 'fused_rows':
                return ops.is_step_rows(eng.params, run.ws, eng.net_handle, 1, 0, n, run.e_obs, pv, prior, h, c, n, rows, None, 3, 0)
            return ops.is_step(eng.params, run.ws, eng.net_handle, 1, 0, n, run.e_obs, pv, prior, h, c, n, None, 3, 0)
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        reps = 10
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        ev[0].record()
        for r in range(reps):
            call()
            ev[r + 1].record()
        torch.cuda.synchronize()
        ms = sorted(ev[r].elapsed_time(ev[r + 1]) for r in range(reps))
        med = ms[len(ms) // 2]
        print(json.dumps({'n': n, 'H': H, 'mode': mode, 'ms_median': round(med, 4), 'ms_min': round(ms[0], 4),
                          'particles_per_s': round(n / med * 1e3), 'tflops_algorithmic': round(flops_alg / med / 1e9, 2),
                          'frac_fp32_mfma_peak': round(flops_alg / med / 1e9 / 157.3, 3),
                          'tflops_executed': round((flops_exe if mode != 'chain' else flops_alg) / med / 1e9, 2),
                          'depth': DEPTH, 'state_bytes_per_particle': 16 * H * DEPTH, 'state_GBps': round(n * 16 * H * DEPTH / med / 1e6, 1)}), flush=True)
