"""Property tests (hypothesis) of the two host-side packers behind the C ABI - pp_pack_ragged (Batch.__init__ +
the step-major layout, pyprob/nn/dataset.py:21-37) and pp_pack_indexed (minibatches straight from memory-mapped dataset
columns) - against the numpy statement of the layout, on arbitrary ragged shapes: single traces, one-statement batches,
long tails, repeated and out-of-order trace ids, several shards, zero observation width. No device involved."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from pyprob_amd.packed import PackedBatch

FIELDS = ('obs', 'value', 'prior', 'addr', 'prev_row', 'trace', 'n_active', 'row_off', 'grp_rows', 'grp_off', 'nxt_rows',
          'nxt_off', 'order', 'src_row', 'cur_counts', 'prev_counts')


@st.composite
def ragged_batches(draw):
    B = draw(st.integers(1, 96))
    max_len = draw(st.sampled_from([1, 2, 3, 7, 20]))
    n_addr = draw(st.integers(1, 9))
    pw = draw(st.integers(1, 3))
    W = draw(st.integers(0, 3))
    seed = draw(st.integers(0, 2 ** 31 - 1))
    rng = np.random.default_rng(seed)
    # a few long traces among short ones (the ragged tail the LSTM tail kernels exist for), or uniform lengths
    lens = rng.integers(1, max_len + 1, B) if draw(st.booleans()) else np.full(B, max_len)
    if draw(st.booleans()):
        lens[rng.integers(0, B)] = max_len + draw(st.integers(0, 12))
    R = int(lens.sum())
    return dict(lens=lens.astype(np.int32), ids=rng.integers(0, n_addr, R), vals=rng.normal(size=R).astype(np.float32),
                prior=rng.normal(size=(R, pw)).astype(np.float32), obs=rng.normal(size=(B, W)).astype(np.float32), n_addr=n_addr)


@settings(derandomize=True, max_examples=120, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(ragged_batches())
def test_pack_ragged_equals_the_numpy_layout(b):
    got = PackedBatch.from_ragged(b['lens'], b['ids'], b['vals'], b['prior'], b['obs'], b['n_addr'])
    want = PackedBatch.from_ragged_numpy(b['lens'], b['ids'], b['vals'], b['prior'], b['obs'], b['n_addr'])
    for n in FIELDS:
        assert np.array_equal(getattr(got, n), getattr(want, n)), n
    assert got.t_max == want.t_max == int(b['lens'].max()) and got.n_rows == want.n_rows == int(b['lens'].sum())
    # the layout's own invariants (step-major: step t holds the traces still alive, longest first)
    assert np.array_equal(got.n_active, [(b['lens'] > t).sum() for t in range(got.t_max)])
    assert np.array_equal(got.row_off, np.concatenate([[0], np.cumsum(got.n_active)]))
    assert sorted(got.src_row.tolist()) == list(range(got.n_rows))
    order = np.asarray(got.order)
    assert sorted(order.tolist()) == list(range(len(b['lens']))) and np.all(np.diff(b['lens'][order]) <= 0)
    assert int(got.cur_counts.sum()) == got.n_rows and int(got.prev_counts.sum()) == got.n_rows - len(b['lens'])


@st.composite
def dataset_and_picks(draw):
    n = draw(st.integers(2, 120))
    n_shards = draw(st.integers(1, 3))
    n_addr = draw(st.integers(1, 6))
    seed = draw(st.integers(0, 2 ** 31 - 1))
    rng = np.random.default_rng(seed)
    shards = []
    for s in range(n_shards):
        m = n if s == 0 else int(rng.integers(1, n + 1))
        lens = rng.integers(1, 6, m)
        R = int(lens.sum())
        shards.append((lens, rng.integers(0, n_addr, R), rng.normal(size=R).astype(np.float32),
                       np.stack([rng.normal(size=R), rng.uniform(0.5, 2.0, R)], 1).astype(np.float32),
                       rng.normal(size=(m, 2)).astype(np.float32)))
    total = sum(len(s[0]) for s in shards)
    k = draw(st.integers(1, 64))
    picks = rng.integers(0, total, k) if draw(st.booleans()) else rng.permutation(total)[:k]      # repeats allowed
    return dict(shards=shards, picks=np.asarray(picks, np.int64), n_addr=n_addr)


class _Spec:
    def __init__(self, table):
        self.addresses = table
        self.address_id = {a[0]: i for i, a in enumerate(table)}
        self.obs = [('obs0', 1, 4, 8), ('obs1', 1, 4, 8)]


@settings(derandomize=True, max_examples=60, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(dataset_and_picks())
def test_pack_indexed_equals_gather_then_pack(d):
    from pyprob_amd.dataset import PackedTraceDataset, _MemoryShard
    table = [('addr_%d' % a, 'Normal', None) for a in range(d['n_addr'])]
    ds = PackedTraceDataset([_MemoryShard(['obs0', 'obs1'], [1, 1], lens, table, ids, vals, prior, obs, None)
                             for lens, ids, vals, prior, obs in d['shards']])
    spec = _Spec(table)
    got = ds.batch(d['picks'], spec)
    lens, ids, vals, prior, obs = ds.gather(d['picks'])[:5]
    amap = np.asarray([spec.address_id[ds.addresses[a][0]] for a in range(len(ds.addresses))])
    want = PackedBatch.from_ragged_numpy(lens, amap[ids], vals, prior, obs, len(table))
    for n in FIELDS:
        assert np.array_equal(getattr(got, n), getattr(want, n)), n
    with pytest.raises(IndexError):
        ds.batch([len(ds)], spec)
    with pytest.raises(ValueError):
        ds.batch([], spec)
