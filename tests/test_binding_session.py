"""The binding's session replay on the CPU (`-m "not gpu"`): `_HipNetworkMixin` over the oracle-backed operators
(tests/oracle_ops.py), driven with the minibatches recorded from the stock reference (tests/golden/make_session.py). This is
the SAME check the MI355X runs in tests/test_gpu_binding_session.py - here it validates the fixtures and the replay logic, and
runs where pyprob is absent too (unlike tests/test_binding_reference.py, which needs the live reference)."""
import pytest

import oracle_ops
from session_checks import check_grad_none_set, check_infer_steps, check_pickle_roundtrip, check_training_session, load_session


def _factory(spec, device):
    return oracle_ops.CpuBufferEngine(spec)


@pytest.mark.parametrize('case', ['gum', 'gumm', 'ffcat', 'gumm2'])
def test_recorded_training_session_through_the_mixin(case):
    net, meta, arrays = check_training_session(case, 'cpu', _factory)
    clone = check_pickle_roundtrip(net, meta, arrays)
    final = load_session(case)[3]
    assert check_infer_steps(clone, meta, arrays, final) >= 24


def test_grad_none_for_the_addresses_a_minibatch_does_not_visit():
    visited, known = check_grad_none_set('gumm', 'cpu', _factory)
    assert visited < known
