"""HBM rate of the optimizer passes over the flat buffers (pp_adam_step, pp_sgd_step, pp_larc_scale):
python tools/optim_bench.py [lstm_dim ...]   ->  one line per kernel: us per call, algorithmic bytes, GB/s, fraction of 8 TB/s.
Algorithmic bytes per (padded) parameter: Adam 16 read + 12 written (+4 with the fused zero_grad), SGD with momentum
12 + 8 (+4), LARC 8 (norms) + 8 read + 4 written (rescaling)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyprob_amd.engine import ICEngine  # noqa: E402
from pyprob_amd.spec import NetSpec  # noqa: E402


def timed(fn, reps=200):
    for _ in range(10):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


def main():
    for H in [int(x) for x in sys.argv[1:]] or [512, 1024, 2048]:
        spec = NetSpec({'obs0': {'dim': 32}, 'obs1': {'dim': 32}}, lstm_dim=H)
        spec.add_address('mu', 'Normal')
        eng = ICEngine(spec, device='cuda:0', seed=0)
        n = spec.n_params
        eng.active.fill_(1.0)
        eng.grads.normal_()
        eng.moments_written()           # every chunk is stepped (no idle shortcut for all-zero gradient chunks)
        rows = [('adam_step', lambda: eng.adam_step(1e-3, zero_grads=False), 28.0),
                ('sgd_step (momentum)', lambda: eng.sgd_step(1e-3, 0.9, True, 0.0, zero_grads=False), 20.0),
                ('larc_scale (3 launches)', lambda: eng.larc_scale(1e-3, 1e-5), 28.0)]
        for name, fn, bytes_per in rows:
            us = timed(fn)
            gbs = bytes_per * n / us / 1e3
            print('H=%d (%d padded parameters)  %-26s %7.2f us  %6.1f MB  %7.0f GB/s  %.2f of 8 TB/s'
                  % (H, n, name, us, bytes_per * n / 1e6, gbs, gbs / 8000.0))


if __name__ == '__main__':
    main()
