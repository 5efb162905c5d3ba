import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, 'tests', 'golden')

# tests/test_binding_reference.py runs the binding under the LIVE reference: it exists in the build container only (the Python
# reference may not travel to the GPU box in any form) - where it is absent the module is not collected at all (a module-level
# skip would show up as a "skipped" GPU test)
collect_ignore = [] if os.path.isdir('/root/reference/pyprob') else ['test_binding_reference.py']


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
