"""the log-weights of 10^4 particles the reference sampled and scored with the golden networks
(tests/golden/make_is_10k.py; weights from -2.7 down to -152), re-computed on the device in lock-step groups of thousands of
particles - prior log_prob, proposal log_prob, observed likelihood terms and their fp32 accumulation through pp_is_step /
pp_logweight_terms - to the 1e-4 BASELINE.json's north_star states. (The scoring loop is the one of
tests/test_gpu_logweight.py::test_batched_log_weights_equal_per_particle.)"""
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_golden
from helpers import engine_from_golden
from test_gpu_logweight import _score_in_groups

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')


@pytest.mark.parametrize('case', ['gum', 'gumm'])
def test_ten_thousand_reference_particles(case):
    meta, params, batch, loss, isr = load_golden(case)
    big = dict(np.load(os.path.join(GOLDEN, case + '_is10k.npz')))
    assert len(big['lw']) == 10000
    _score_in_groups(case, engine_from_golden(meta, params), big, [str(a) for a in big['addresses']])
