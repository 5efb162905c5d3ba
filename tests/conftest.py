import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(case):
    """Golden vectors recorded from the reference by tests/golden/make_golden.py."""
    with open(os.path.join(GOLDEN, case + '_meta.json')) as f:
        meta = json.load(f)
    net = np.load(os.path.join(GOLDEN, case + '_net.npz'))
    params = {n: net['p%d' % i] for i, n in enumerate(meta['state_dict_names'])}
    batch = dict(np.load(os.path.join(GOLDEN, case + '_batch.npz')))
    loss = dict(np.load(os.path.join(GOLDEN, case + '_loss.npz')))
    isr = dict(np.load(os.path.join(GOLDEN, case + '_is.npz')))
    return meta, params, batch, loss, isr


@pytest.fixture(params=['gum', 'gumm', 'cat', 'poi', 'ff', 'ffc', 'ber', 'gumm2', 'gumd'])
def golden(request):
    return (request.param,) + load_golden(request.param)
