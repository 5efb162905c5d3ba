"""Packed on-disk trace dataset (pyprob_amd/dataset.py): format round trip, the reference's sort order
and sampler semantics, shard concatenation, minibatch packing. CPU only."""
import json
import os

import numpy as np
import pytest

from pyprob_amd.dataset import PackedTraceDataset, PackedTraceWriter, fnv1a64, trace_type_hash
from pyprob_amd.packed import PackedBatch
from pyprob_amd.parallel import DistributedTraceBatchSampler


def ragged(n, seed, n_addr=5, max_len=4):
    """Synthetic ragged traces: a trace of length L visits addresses start, start+1, ... (mod n_addr)."""
    rng = np.random.default_rng(seed)
    lens = rng.integers(1, max_len + 1, n)
    start = rng.integers(0, 2, n)
    ids = np.concatenate([(s + np.arange(L)) % n_addr for s, L in zip(start, lens)])
    R = int(lens.sum())
    value = rng.normal(size=R).astype(np.float32)
    prior = np.stack([rng.normal(size=R), rng.uniform(0.5, 2.0, R)], 1).astype(np.float32)
    obs = rng.normal(size=(n, 2)).astype(np.float32)
    table = [('addr_%d' % a, 'Normal' if a % 2 == 0 else 'Uniform', None) for a in range(n_addr)]
    return lens, table, ids, value, prior, obs


def write(path, n, seed, **kw):
    cols = ragged(n, seed, **kw)
    with PackedTraceWriter(path, ['obs0', 'obs1'], [1, 1]) as w:
        w.add_columns(*cols)
    return cols


def test_fnv_is_process_independent():
    assert fnv1a64(b'') == 0xcbf29ce484222325
    assert fnv1a64(b'a') == 0xaf63dc4c8601ec8c           # published FNV-1a test vector
    assert trace_type_hash(['x', 'y']) != trace_type_hash(['xy'])


def test_round_trip_and_sort_order(tmp_path):
    lens, table, ids, value, prior, obs = write(str(tmp_path / 's0'), 500, 1)
    ds = PackedTraceDataset(str(tmp_path / 's0'))
    assert len(ds) == 500 and ds.obs_width == 2
    meta = json.load(open(tmp_path / 's0' / 'meta.json'))
    assert meta['n_rows'] == int(lens.sum()) and meta['version'] == 1
    # sorted by (length, type hash), like OfflineDataset's sorted index (dataset.py:217-259)
    hashes = np.asarray([h for h, _ in ds.trace_types], np.uint64)[ds.trace_type]
    key = list(zip(ds.trace_len.tolist(), hashes.tolist()))
    assert key == sorted(key)
    assert np.array_equal(ds.sorted_indices(), np.arange(500))
    # every stored trace is one of the written traces (multiset equality over (obs, values, addresses))
    off = np.concatenate([[0], np.cumsum(lens)])
    want = sorted((tuple(obs[i]), tuple(value[off[i]:off[i + 1]]), tuple(table[a][0] for a in ids[off[i]:off[i + 1]]))
                  for i in range(500))
    l2, a2, v2, p2, o2 = ds.gather(np.arange(500))
    off2 = np.concatenate([[0], np.cumsum(l2)])
    got = sorted((tuple(o2[i]), tuple(v2[off2[i]:off2[i + 1]]), tuple(ds.addresses[a][0] for a in a2[off2[i]:off2[i + 1]]))
                 for i in range(500))
    assert got == want
    # gather honours the requested order and duplicates
    sel = np.array([17, 3, 17, 499, 0])
    l3, a3, v3, p3, o3 = ds.gather(sel)
    assert np.array_equal(l3, l2[sel]) and np.array_equal(o3, o2[sel])
    assert np.array_equal(v3[:l3[0]], v2[off2[17]:off2[18]])
    assert np.array_equal(p3[l3[0]:l3[0] + l3[1]], p2[off2[3]:off2[4]])


def test_shards_concatenate_with_different_address_tables(tmp_path):
    write(str(tmp_path / 'a'), 200, 2, n_addr=3)
    write(str(tmp_path / 'b'), 300, 3, n_addr=6)
    ds = PackedTraceDataset(str(tmp_path))
    assert len(ds) == 500 and len(ds.addresses) == 6
    order = ds.sorted_indices()
    assert sorted(order.tolist()) == list(range(500))
    assert np.all(np.diff(ds.trace_len[order]) >= 0)
    lens, addr, value, prior, obs = ds.gather(order[:64])
    assert addr.max() < 6 and len(value) == lens.sum()
    with pytest.raises(IndexError):
        ds.gather([500])
    with pytest.raises(ValueError):
        ds.gather([])


def test_pruned_trace_view_and_writer_from_traces(tmp_path):
    write(str(tmp_path / 's'), 50, 4)
    ds = PackedTraceDataset(str(tmp_path / 's'))
    tr = ds[7]
    lens, addr, value, prior, obs = ds.gather([7])
    assert tr.length_controlled == lens[0] == len(tr.variables_controlled)
    assert [v.address for v in tr.variables_controlled] == [ds.addresses[a][0] for a in addr]
    assert np.allclose([float(v.value) for v in tr.variables_controlled], value)
    assert np.allclose([float(tr.named_variables[n].value) for n in ('obs0', 'obs1')], obs[0])
    d0 = tr.variables_controlled[0].distribution
    assert d0.name == ds.addresses[addr[0]][1]
    # Trace objects -> writer -> identical columns
    with PackedTraceWriter(str(tmp_path / 't'), ['obs0', 'obs1']) as w:
        for i in range(50):
            w.add_trace(ds[i])
    ds2 = PackedTraceDataset(str(tmp_path / 't'))
    g1, g2 = ds.gather(np.arange(50)), ds2.gather(np.arange(50))
    for k, (a, b) in enumerate(zip(g1, g2)):
        if k == 1:      # address ids are per-file (first appearance): compare the address strings
            assert [ds.addresses[i][0] for i in a] == [ds2.addresses[i][0] for i in b]
        elif a.dtype.kind == 'f':
            assert np.allclose(a, b, rtol=1e-6, atol=1e-6)
        else:
            assert np.array_equal(a, b)
    with pytest.raises(ValueError):
        PackedTraceWriter(str(tmp_path / 'e'), ['obs0']).close()


def test_sampler_partitions_like_the_reference(tmp_path):
    write(str(tmp_path / 's'), 1000, 5)
    ds = PackedTraceDataset(str(tmp_path / 's'))
    world, bs = 2, 16
    seen = []
    for rank in range(world):
        s = ds.sampler(bs, rank, world, num_buckets=4)
        ref = DistributedTraceBatchSampler(ds.sorted_indices().tolist(), bs, rank, world, 4)
        np.random.seed(0)
        got = [list(b) for b in s]
        np.random.seed(0)
        assert got == [list(b) for b in ref]
        for b in got:
            assert len(b) == bs
            assert ds.trace_len[b].max() - ds.trace_len[b].min() <= 1      # sorted by length: near-uniform minibatches
        seen += [i for b in got for i in b]
    assert len(seen) == len(set(seen))                                      # ranks take disjoint minibatches


class _Spec:
    def __init__(self, addresses, obs=(('obs0', 1, 4, 8), ('obs1', 1, 4, 8))):
        self.addresses = addresses
        self.address_id = {a[0]: i for i, a in enumerate(addresses)}
        self.obs = list(obs)          # (name, input width, hidden, output) like NetSpec.obs


def test_batch_matches_from_ragged_and_loader_prefetches(tmp_path):
    write(str(tmp_path / 's'), 400, 6)
    ds = PackedTraceDataset(str(tmp_path / 's'))
    spec = _Spec(list(reversed(ds.addresses)))          # the network numbers addresses differently from the file
    ids = ds.sorted_indices()[100:164]
    pb = ds.batch(ids, spec)
    lens, addr, value, prior, obs = ds.gather(ids)
    remap = np.asarray([spec.address_id[a[0]] for a in ds.addresses])
    ref = PackedBatch.from_ragged(lens, remap[addr], value, prior, obs, len(spec.addresses))
    for name in ('obs', 'value', 'prior', 'addr', 'prev_row', 'trace', 'n_active', 'row_off', 'grp_rows', 'grp_off'):
        assert np.array_equal(getattr(pb, name), getattr(ref, name)), name
    assert ds.addresses_of(ids) and all(a in ds.addresses for a in ds.addresses_of(ids))

    class Host(PackedBatch):
        pass
    # loader: same minibatches as the sampler, in order, then stops after `epochs`
    import pyprob_amd.packed as P
    uploaded = []
    orig = P.PackedBatch.to
    P.PackedBatch.to = lambda self, device: uploaded.append(self.n_traces) or self
    try:
        np.random.seed(1)
        got = [b for b in ds.loader(spec, 32, 'cpu', epochs=2, prefetch=2, shuffle_batches=False)]
    finally:
        P.PackedBatch.to = orig
    assert len(got) == 2 * (400 // 32) and uploaded == [32] * len(got)


def test_vectorised_prior_traces_match_the_per_trace_generator():
    """Model.prior_traces_packed: prior traces generated in lock step, one forward() per control-flow
    path, have the per-trace generator's addresses (instance counters included), the rejection-loop structure and the
    right marginals. CPU only."""
    import sys
    import torch
    sys.path.insert(0, os.path.dirname(__file__))
    from models import GaussianWithUnknownMean, GaussianWithUnknownMeanMarsagliaLockStep
    from pyprob_amd.dataset import VectorisedOnlineDataset
    from pyprob_amd.state import TraceMode
    torch.manual_seed(3)
    model = GaussianWithUnknownMeanMarsagliaLockStep()
    lens, table, ids, vals, prior, obs = model.prior_traces_packed(30000, ['obs0', 'obs1'])
    assert len(lens) == 30000 and lens.sum() == len(vals) == len(ids) and obs.shape == (30000, 2)
    assert np.all(lens % 2 == 0) and np.all(prior == np.array([-1.0, 1.0], np.float32))
    # per-trace generator: same address strings for the same statements
    gen = model._trace_generator(trace_mode=TraceMode.PRIOR_FOR_INFERENCE_NETWORK)
    seen = {}
    for _ in range(300):
        tr = next(gen)
        seen[tr.length_controlled] = [v.address for v in tr.variables_controlled]
    names = [t[0] for t in table]
    off = np.concatenate([[0], np.cumsum(lens)])
    for L, addrs in seen.items():
        i = int(np.nonzero(lens == L)[0][0])
        assert [names[a] for a in ids[off[i]:off[i + 1]]] == addrs
    # rejection structure: only the last (x, y) pair of a trace lies inside the unit disc
    for i in range(0, 30000, 37):
        s2 = (vals[off[i]:off[i + 1]].reshape(-1, 2) ** 2).sum(1)
        assert s2[-1] < 1 and np.all(s2[:-1] >= 1)
    # P(length = 2k) = (pi/4)(1 - pi/4)^(k-1); observations ~ N(mu, 2) with mu ~ N(1, 5)
    p1 = np.mean(lens == 2)
    assert abs(p1 - np.pi / 4) < 0.01
    assert abs(obs.mean() - 1.0) < 0.05 and abs(obs.var() - 7.0) < 0.3
    # straight-line program: one path, exact prior moments
    lens, table, ids, vals, prior, obs = GaussianWithUnknownMean().prior_traces_packed(200000, ['obs0', 'obs1'])
    assert len(table) == 1 and np.all(lens == 1)
    assert abs(vals.mean() - 1.0) < 0.02 and abs(vals.std() - np.sqrt(5)) < 0.02
    assert np.allclose(prior[0], [1.0, np.sqrt(5)], atol=1e-6)
    # served as an online dataset: fresh traces on refresh, same interface as the on-disk dataset
    ds = VectorisedOnlineDataset(model, ['obs0', 'obs1'], chunk_traces=4096)
    first = ds.gather(np.arange(16))[2].copy()
    ds.refresh()
    assert ds.generated == 8192 and not np.array_equal(first, ds.gather(np.arange(16))[2])
    assert ds[0].length_controlled == ds.trace_len[0]
    assert len([b for b in ds.sampler(256)]) == 4096 // 256


def test_known_trace_types_equal_detected_types():
    """add_columns(trace_types=...) (the lock-step generator knows one type per control-flow path) builds the same
    dataset as the row-wise unique detection."""
    lens, table, ids, value, prior, obs = ragged(400, 9)
    off = np.concatenate([[0], np.cumsum(lens)])
    seqs, index, type_of = [], {}, np.empty(400, np.int64)
    for i in range(400):
        key = tuple(ids[off[i]:off[i + 1]].tolist())
        if key not in index:
            index[key] = len(seqs)
            seqs.append(list(key))
        type_of[i] = index[key]
    a = PackedTraceDataset.from_columns(['obs0', 'obs1'], [1, 1], lens, table, ids, value, prior, obs)
    b = PackedTraceDataset.from_columns(['obs0', 'obs1'], [1, 1], lens, table, ids, value, prior, obs,
                                        trace_types=(type_of, seqs))
    ha = np.asarray([h for h, _ in a.trace_types], np.uint64)[a.trace_type]
    hb = np.asarray([h for h, _ in b.trace_types], np.uint64)[b.trace_type]
    assert np.array_equal(ha, hb) and np.array_equal(a.trace_len, b.trace_len)
    for x, y in zip(a.gather(np.arange(400)), b.gather(np.arange(400))):
        assert np.array_equal(np.asarray(x), np.asarray(y))


def test_prior_inflation_draws_wider_values_and_keeps_the_prior_parameters():
    """PriorInflation.ENABLED (pyprob/state.py:87-93, 280-288): Normal values are drawn with 3x the standard deviation
    and Categorical values uniformly, the prior parameters the heads see stay the program's; the per-trace generator
    carries the correcting importance weight. Both generators (per trace and lock step)."""
    import sys
    import torch
    sys.path.insert(0, os.path.dirname(__file__))
    from models import CategoricalThenNormal, GaussianWithUnknownMean
    from pyprob_amd.state import PriorInflation, TraceMode
    torch.manual_seed(11)
    gum = GaussianWithUnknownMean()
    for inflation, sd in ((PriorInflation.DISABLED, np.sqrt(5.0)), (PriorInflation.ENABLED, 3 * np.sqrt(5.0))):
        lens, table, ids, vals, prior, obs = gum.prior_traces_packed(40000, ['obs0', 'obs1'], prior_inflation=inflation)
        assert abs(vals.std() - sd) < 0.03 * sd and abs(vals.mean() - 1.0) < 0.1
        assert np.allclose(prior, np.array([1.0, np.sqrt(5.0)], np.float32))
    gen = gum._trace_generator(trace_mode=TraceMode.PRIOR_FOR_INFERENCE_NETWORK, prior_inflation=PriorInflation.ENABLED)
    tr = [next(gen) for _ in range(3000)]
    v = np.asarray([float(t.variables_controlled[0].value) for t in tr])
    assert abs(v.std() - 3 * np.sqrt(5.0)) < 0.5
    var = tr[0].variables_controlled[0]
    want = float(var.distribution.log_prob(var.value)) - float(
        torch.distributions.Normal(1.0, 3 * np.sqrt(5.0)).log_prob(var.value))
    assert abs(var.log_importance_weight - want) < 1e-5 and abs(float(var.distribution.stddev) - np.sqrt(5.0)) < 1e-6
    # importance weights undo the inflation: the weighted prior mean/variance of mu are the program's
    lw = np.asarray([t.log_importance_weight for t in tr])
    w = np.exp(lw - lw.max())
    w /= w.sum()
    m = (w * v).sum()
    assert abs(m - 1.0) < 0.25 and abs((w * (v - m) ** 2).sum() - 5.0) < 0.8
    # Categorical: uniform draws
    cat = CategoricalThenNormal()
    lens, table, ids, vals, prior, obs = cat.prior_traces_packed(30000, ['obs0', 'obs1'],
                                                                 prior_inflation=PriorInflation.ENABLED)
    c = vals.reshape(-1, 2)[:, 0]
    assert np.allclose(np.bincount(c.astype(int), minlength=3) / 30000.0, 1 / 3, atol=0.015)


def test_online_dataset_prefetches_the_next_chunk_in_a_worker_thread():
    """VectorisedOnlineDataset: refresh() serves the chunk a worker thread prepared (start_prefetch), or generates one;
    a failing generator surfaces in refresh(); wait_prefetch() joins without swapping."""
    import sys
    import torch
    sys.path.insert(0, os.path.dirname(__file__))
    from models import GaussianWithUnknownMeanMarsagliaLockStep
    from pyprob_amd.dataset import VectorisedOnlineDataset
    torch.manual_seed(2)
    model = GaussianWithUnknownMeanMarsagliaLockStep()
    ds = VectorisedOnlineDataset(model, ['obs0', 'obs1'], chunk_traces=2000)
    first = ds.obs_view = np.array(ds.gather(np.arange(5))[4])
    assert len(ds.trace_len) == 2000 and ds.generated == 2000
    ds.start_prefetch()
    ds.start_prefetch()                    # (idempotent while a chunk is pending)
    ds.wait_prefetch()
    assert ds.generated == 2000 and np.array_equal(np.array(ds.gather(np.arange(5))[4]), first)   # not swapped yet
    ds.refresh()
    assert ds.generated == 4000 and not np.array_equal(np.array(ds.gather(np.arange(5))[4]), first)
    ds.refresh()                           # nothing prepared: generated on the spot
    assert ds.generated == 6000 and len(ds.trace_len) == 2000

    class Broken(GaussianWithUnknownMeanMarsagliaLockStep):
        calls = 0

        def forward(self):
            Broken.calls += 1
            if Broken.calls > 40:
                raise RuntimeError('generator failed')
            return super().forward()
    bad = VectorisedOnlineDataset(Broken(), ['obs0', 'obs1'], chunk_traces=500)
    Broken.calls = 1000
    bad.start_prefetch()
    with pytest.raises(RuntimeError, match='generator failed'):
        bad.refresh()


def test_validation_loss_averages_the_minibatch_losses(tmp_path):
    """InferenceNetwork._validation_loss (inference_network.py:538-543): mean of the forward-only loss over the
    minibatches of a packed validation set; minibatches with unknown addresses are skipped. (Engine faked: CPU.)"""
    import torch
    from pyprob_amd.nn import InferenceNetworkLSTM
    write(str(tmp_path / 'v'), 300, 4)
    ds = PackedTraceDataset(str(tmp_path / 'v'))

    class FakeEngine:
        device = 'cpu'
        spec = _Spec(list(ds.addresses))
        seen = []

        def loss(self, pb):
            self.seen.append(pb.n_traces)
            return torch.tensor([float(pb.n_rows) / pb.n_traces])
    net = InferenceNetworkLSTM(observe_embeddings={'obs0': {'dim': 8}, 'obs1': {'dim': 8}})
    net._engine = FakeEngine()
    net._obs_names = ['obs0', 'obs1']
    got = net._validation_loss(ds, 64)
    assert FakeEngine.seen == [64, 64, 64, 64]                       # 300 // 64 full minibatches, sorted order
    lens = ds.trace_len[ds.sorted_indices()[:256]].reshape(4, 64)
    assert abs(got - float(np.mean(lens.mean(1)))) < 1e-6
    net._engine.spec = _Spec(list(ds.addresses)[:1])                # the network knows only the first address
    assert np.isnan(net._validation_loss(ds, 64)) or net._validation_loss(ds, 64) > 0
    # a dataset whose observable columns are not the network's (order, names or widths) is refused, not mis-read
    net._obs_names = ['obs1', 'obs0']
    with pytest.raises(ValueError, match='do not match the observe embeddings'):
        net._validation_loss(ds, 64)


def test_save_dataset_detects_observed_variables_only(tmp_path):
    """Auto-detected observables are the variables recorded by observe(), not every NAMED variable: a named latent
    sample must not become an input column of the inference network."""
    import pyprob_amd as pyprob
    from pyprob_amd.distributions import Normal
    from pyprob_amd.model import Model

    class NamedLatent(Model):
        def forward(self):
            mu = pyprob.sample(Normal(0., 1.), name='mu')
            pyprob.observe(Normal(mu, 1.), name='obs')
            return mu
    NamedLatent().save_dataset(str(tmp_path / 'd'), 50, 50)
    ds = PackedTraceDataset(str(tmp_path / 'd'))
    assert ds.obs_names == ['obs'] and ds.obs_width == 1


def test_bernoulli_programs_pack_from_a_dataset(tmp_path):
    """A program with a Bernoulli proposal (proposal_bernoulli_bernoulli.py): the rows of its head carry (n, sum of
    values) of their sub-batch step. The dataset route (columns -> pp_pack_indexed + the step statistics) gives the same
    packed minibatch as the Trace route (Batch -> pack_traces + bernoulli_group_stats)."""
    from models import BernoulliThenNormal
    from pyprob_amd.packed import pack_traces
    from pyprob_amd.spec import NetSpec
    torch = pytest.importorskip('torch')
    torch.manual_seed(3)
    BernoulliThenNormal().save_dataset(str(tmp_path / 'b'), 300, 150)
    ds = PackedTraceDataset(str(tmp_path / 'b'))
    spec = NetSpec({'obs0': {'dim': 8}, 'obs1': {'dim': 8}}, lstm_dim=16)
    for a, d, nc in ds.addresses:
        spec.add_address(a, d, nc)
    assert any(a.dist_name == 'Bernoulli' for a in spec.addresses)
    ids = ds.sorted_indices()[40:140]
    got = ds.batch(ids, spec)
    ref = pack_traces([ds[int(i)] for i in ids], spec, ['obs0', 'obs1'])
    np.testing.assert_array_equal(got.addr, ref.addr)
    np.testing.assert_array_equal(got.value, ref.value)
    np.testing.assert_allclose(got.prior, ref.prior)
    b = spec.address_id[[a.address for a in spec.addresses if a.dist_name == 'Bernoulli'][0]]
    rows = got.addr == b
    assert np.all(got.prior[rows, 0] == rows.sum()) and np.all(got.prior[rows, 1] == got.value[rows].sum())


def test_columnar_dataset_blocks_and_refill():
    """packed.ColumnarDataset: a minibatch is one contiguous block [obs | value | prior]; refill() writes a new chunk into the SAME
    storage (the PackedBatch objects of the blocks - raw pointers - stay valid, nn.optimize's resident chunks) and refuses a chunk
    of another shape."""
    import torch
    from pyprob_amd.packed import ColumnarDataset
    g = torch.Generator().manual_seed(3)
    n, B, w = 70, 16, 2
    obs, val, pri = torch.randn(n, w, generator=g), torch.randn(n, generator=g), torch.randn(n, 2, generator=g)
    cd = ColumnarDataset(obs, val, pri, B)
    assert cd.n_batches == 4 and cd.blocks.shape == (4, B * (w + 3))
    for i in range(4):
        o, v, p = cd.columns(i)
        assert torch.equal(o, obs[i * B:(i + 1) * B]) and torch.equal(v, val[i * B:(i + 1) * B]) and torch.equal(p, pri[i * B:(i + 1) * B])
        assert o.data_ptr() == cd.blocks[i].data_ptr()                      # views, not copies
    ptr = cd.blocks.data_ptr()
    obs2, val2, pri2 = obs + 1, val * 2, pri - 1
    assert cd.refill(obs2, val2, pri2) and cd.blocks.data_ptr() == ptr
    o, v, p = cd.columns(2)
    assert torch.equal(o, obs2[2 * B:3 * B]) and torch.equal(v, val2[2 * B:3 * B]) and torch.equal(p, pri2[2 * B:3 * B])
    assert not cd.refill(obs2[:40], val2[:40], pri2[:40])                     # two blocks instead of four: a new dataset is needed
    assert torch.equal(cd.columns(3)[1], val2[3 * B:4 * B])                   # ... and nothing was touched
