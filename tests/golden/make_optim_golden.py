#!/usr/bin/env python3
"""Golden optimizer trajectories under tests/golden/optim_steps.npz, recorded by RUNNING the optimizers the reference
constructs in InferenceNetwork._create_optimizer (pyprob/nn/inference_network.py:343-355): torch.optim.Adam,
torch.optim.SGD(momentum, nesterov=True), each alone and inside the reference's LARC wrapper
(pyprob/nn/optimizer_larc.py). Build container only (needs /root/reference):

    python tests/golden/make_optim_golden.py

Per case <opt>_<wd index>: three float64 tensors (one of them all-zero initially: LARC's epsilon branch), six steps with
recorded gradients (step 2 has gradients 10x larger: LARC's clipping engages), tensor 1 has no gradient at step 1
(grad is None: no update, no decay, no state). Recorded: initial parameters, gradients, presence, parameters after every
step."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(REPO, 'oracle', 'refstubs'))
sys.path.insert(1, '/root/reference')

import numpy as np  # noqa: E402
import torch  # noqa: E402
from pyprob.nn.optimizer_larc import LARC  # noqa: E402

SHAPES = [(7, 5), (33,), (4, 4)]
LR, MOMENTUM, STEPS = 0.05, 0.9, 6
WDS = (0.0, 1e-3)


def main():
    rng = np.random.default_rng(2024)
    out = dict(lr=LR, momentum=MOMENTUM, weight_decays=np.asarray(WDS))
    for name in ('ADAM', 'SGD', 'ADAM_LARC', 'SGD_LARC'):
        for w, wd in enumerate(WDS):
            ps = [torch.nn.Parameter(torch.tensor(rng.normal(size=s) * (0.0 if i == 2 else 1.0), dtype=torch.float64))
                  for i, s in enumerate(SHAPES)]
            if name.startswith('ADAM'):
                opt = torch.optim.Adam(ps, lr=LR, weight_decay=wd)                                       # :348
            else:
                opt = torch.optim.SGD(ps, lr=LR, momentum=MOMENTUM, nesterov=True, weight_decay=wd)     # :350
            if name.endswith('LARC'):
                opt = LARC(opt)                                                                          # :352
            key = '{}_{}'.format(name, w)
            for k, p in enumerate(ps):
                out['{}_p0_{}'.format(key, k)] = p.detach().numpy().copy()
            for it in range(STEPS):
                present = [True, it != 1, True]
                for k, (p, s) in enumerate(zip(ps, SHAPES)):
                    g = rng.normal(size=s) * (10.0 if it == 2 else 1.0)
                    out['{}_g{}_{}'.format(key, it, k)] = g
                    p.grad = torch.tensor(g) if present[k] else None
                out['{}_present{}'.format(key, it)] = np.asarray(present)
                opt.step()
                for k, p in enumerate(ps):
                    out['{}_p{}_{}'.format(key, it + 1, k)] = p.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, 'optim_steps.npz'), **out)
    print('wrote optim_steps.npz with', len(out), 'arrays')


if __name__ == '__main__':
    main()
