"""ParticleTensor's per-call reuse of pure elementwise results (state.py; ADVICE r04 state.py:177): the key separates views
of one storage by shape / strides / offset, a result shared by two expressions is never modified in place, and what one
call keeps is bounded. Host logic (CPU tensors, a stand-in executor)."""
import types

import pytest
import torch

from pyprob_amd import state
from pyprob_amd.state import ParticleTensor


@pytest.fixture
def ls():
    fake = types.SimpleNamespace(memo={}, memo_shared={}, memo_bytes=0, draw=None, width=-1)
    old = state._lock_step
    state._lock_step = fake
    try:
        yield fake
    finally:
        state._lock_step = old


def P(t):
    return ParticleTensor.wrap(t)


def test_views_of_one_storage_do_not_collide(ls):
    x, y = P(torch.arange(6.0)), P(torch.arange(6.0) * 10)
    a = x - y                          # [6]
    b = x.unsqueeze(1) - y             # [6, 6]: same storage address, numel and version as x
    assert a.shape == (6,) and b.shape == (6, 6)
    assert torch.equal(b.as_subclass(torch.Tensor), torch.arange(6.0).unsqueeze(1) - torch.arange(6.0) * 10)
    even, first = x[::2] * 2.0, x[:3] * 2.0
    assert torch.equal(even.as_subclass(torch.Tensor), torch.tensor([0.0, 4.0, 8.0]))
    assert torch.equal(first.as_subclass(torch.Tensor), torch.tensor([0.0, 2.0, 4.0]))
    tail = x[3:] * 2.0                 # same shape and strides as x[:3], another offset
    assert torch.equal(tail.as_subclass(torch.Tensor), torch.tensor([6.0, 8.0, 10.0]))


def test_identical_expressions_are_served_once_and_stay_independent(ls):
    x = P(torch.arange(4.0))
    a = x * x
    b = x * x
    assert a.data_ptr() == b.data_ptr()            # the second product is the first one's result
    a += 1.0                                       # out of place for a shared result: `a` is rebound, `b` keeps the product
    assert torch.equal(b.as_subclass(torch.Tensor), torch.arange(4.0) ** 2)
    assert torch.equal(a.as_subclass(torch.Tensor), torch.arange(4.0) ** 2 + 1.0)
    c = x * x                                      # still valid (nothing wrote into it)
    assert c.data_ptr() == b.data_ptr()
    with pytest.raises(RuntimeError, match='in-place'):
        b.add_(1.0)                                # a method-call form can not be redirected: loud, not silent
    with pytest.raises(RuntimeError, match='in-place'):
        b[0] = 5.0
    with pytest.raises(RuntimeError, match='in-place'):
        b[1:].add_(1.0)                            # a partial VIEW of the shared result has another data_ptr: same storage


def test_a_shared_result_outlives_the_memo(ls, monkeypatch):
    """ADVICE r05 (state.py:248): the shared set is keyed by storage identity and keeps the tensor, so a cleared memo cannot
    hand the address of a shared result to an unrelated tensor (spurious error / silent redirect of its +=)."""
    monkeypatch.setattr(ParticleTensor, 'MEMO_BYTES', 2 * 4 * 64)
    x = P(torch.arange(64.0))
    a = x * x
    b = x * x
    key = b.as_subclass(torch.Tensor).untyped_storage()._cdata
    assert key in ls.memo_shared
    for k in range(8):                             # the memo overflows and is cleared
        _ = x * float(k + 2)
    del a, b
    assert key in ls.memo_shared and ls.memo_shared[key] is not None      # still alive: its address cannot be reused
    fresh = x + 100.0
    fresh += 1.0                                   # an unrelated tensor: in place, no error
    assert torch.equal(fresh.as_subclass(torch.Tensor), torch.arange(64.0) + 101.0)


def test_unshared_results_may_be_modified_in_place(ls):
    x, y = P(torch.arange(4.0)), P(torch.ones(4))
    s = x * x
    s += y * y                                     # (the Marsaglia loop of tests/test_gpu_is_fused.py::MarsagliaInPlace)
    assert torch.equal(s.as_subclass(torch.Tensor), torch.arange(4.0) ** 2 + 1.0)
    t = x * x                                      # the modified product is not handed out as x * x again
    assert torch.equal(t.as_subclass(torch.Tensor), torch.arange(4.0) ** 2)


def test_the_memo_is_bounded(ls, monkeypatch):
    monkeypatch.setattr(ParticleTensor, 'MEMO_BYTES', 3 * 4 * 100)
    x = P(torch.arange(100.0))
    for k in range(10):
        _ = x * float(k)
    assert len(ls.memo) <= 3 and ls.memo_bytes <= 3 * 4 * 100
