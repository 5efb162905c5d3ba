"""CPU tests of the host-side mirror of the trace runtime (pyprob_amd/{state,trace,model,nn,distributions}.py):
trace structure, addresses, sub-batching, packing, IS with prior proposals. No GPU."""
import numpy as np
import pytest
import torch

from models import GaussianWithUnknownMean, GaussianWithUnknownMeanMarsaglia, CategoricalThenNormal
from oracle import ic_oracle as O
from pyprob_amd.nn import Batch
from pyprob_amd.packed import pack_traces
from pyprob_amd.spec import NetSpec
from pyprob_amd.state import InferenceEngine, TraceMode


def _prior_traces(model, n):
    gen = model._trace_generator(trace_mode=TraceMode.PRIOR_FOR_INFERENCE_NETWORK)
    return [next(gen) for _ in range(n)]


def test_gum_trace_structure():
    torch.manual_seed(1)
    traces = _prior_traces(GaussianWithUnknownMean(), 20)
    for t in traces:
        assert t.length_controlled == 1 and len(t.variables_observed) == 2        # reference tests/test_trace.py counts
        assert set(t.named_variables) == {'obs0', 'obs1'}
        v = t.variables_controlled[0]
        assert v.address.endswith('__Normal__1') and '__forward__' in v.address
        assert v.distribution.name == 'Normal'
    assert len({t.variables_controlled[0].address for t in traces}) == 1
    a0 = traces[0].named_variables['obs0'].address
    a1 = traces[0].named_variables['obs1'].address
    assert a0 != a1


def test_gumm_trace_lengths_and_instances():
    torch.manual_seed(2)
    traces = _prior_traces(GaussianWithUnknownMeanMarsaglia(), 400)
    lens = np.array([t.length_controlled for t in traces])
    assert lens.min() == 2 and np.all(lens % 2 == 0)
    assert abs(lens.mean() - 2.546) < 0.25        # reference tests/test_model.py:80: 2.563
    long = [t for t in traces if t.length_controlled >= 4][0]
    addrs = [v.address for v in long.variables_controlled]
    assert addrs[0].endswith('__Uniform__1') and addrs[2].endswith('__Uniform__2')
    assert addrs[0].rsplit('__', 1)[0] == addrs[2].rsplit('__', 1)[0]          # same call site, next instance
    assert addrs[0] != addrs[1]


def test_batch_sub_batching_and_packing_agree_with_oracle():
    torch.manual_seed(3)
    traces = _prior_traces(GaussianWithUnknownMeanMarsaglia(), 64)
    batch = Batch(traces)
    assert batch.size == 64 and abs(batch.mean_length_controlled - np.mean([t.length_controlled for t in traces])) < 1e-12
    spec = NetSpec({'obs0': {'dim': 32}, 'obs1': {'dim': 32}}, lstm_dim=64)
    for t in traces:
        for v in t.variables_controlled:
            spec.add_address(v.address, v.distribution.name)
    pb = pack_traces(traces, spec, ['obs0', 'obs1'])
    # the same grouping the reference's Batch makes (oracle.split_sub_batches restates dataset.py:21-37)
    ids = [spec.address_id[v.address] for t in traces for v in t.variables_controlled]
    subs, _ = O.split_sub_batches([t.length_controlled for t in traces], np.array(ids))
    assert sorted(map(len, subs)) == sorted(map(len, batch.sub_batches))
    assert pb.n_rows == sum(t.length_controlled for t in traces)
    vals = np.array([float(v.value) for t in traces for v in t.variables_controlled], np.float32)
    np.testing.assert_array_equal(pb.value, vals[pb.src_row])
    np.testing.assert_allclose(pb.prior[:, 0], -1.0)
    np.testing.assert_allclose(pb.prior[:, 1], 1.0)
    obs = np.array([[float(t.named_variables[n].value) for n in ('obs0', 'obs1')] for t in traces], np.float32)
    np.testing.assert_array_equal(pb.obs, obs[pb.order])
    with pytest.raises(ValueError):
        Batch([type('T', (), {'length_controlled': 0, 'variables_controlled': []})()])


def test_importance_sampling_with_prior_proposals_gum():
    """IMPORTANCE_SAMPLING engine (no network) on the host: posterior mean 7.25, std sqrt(1/1.2) for obs (8, 9)
    (reference tests/test_inference.py:118-145 thresholds)."""
    torch.manual_seed(4)
    model = GaussianWithUnknownMean()
    post = model.posterior_results(4000, InferenceEngine.IMPORTANCE_SAMPLING, observe={'obs0': 8, 'obs1': 9})
    assert abs(post.mean - 7.25) < 0.75
    assert abs(post.stddev - np.sqrt(1 / 1.2)) < 0.75
    assert 1 < post.effective_sample_size < 4000
    # log weight of one trace = sum of the two likelihood terms (state.py:147-149, trace.py:123-125)
    gen = model._trace_generator(trace_mode=TraceMode.POSTERIOR, observe={'obs0': 8, 'obs1': 9})
    t = next(gen)
    mu = float(t.result)
    ref = O.normal_log_prob(8.0, mu, np.sqrt(2)) + O.normal_log_prob(9.0, mu, np.sqrt(2))
    assert abs(t.log_importance_weight - ref) < 1e-4


def test_posterior_with_network_requires_network():
    model = GaussianWithUnknownMean()
    with pytest.raises(RuntimeError):
        model.posterior_results(10, InferenceEngine.IMPORTANCE_SAMPLING_WITH_INFERENCE_NETWORK, observe={'obs0': 8, 'obs1': 9})


def test_categorical_program_traces():
    torch.manual_seed(5)
    traces = _prior_traces(CategoricalThenNormal(), 10)
    for t in traces:
        names = [v.distribution.name for v in t.variables_controlled]
        assert names == ['Categorical', 'Normal']
        assert t.variables_controlled[0].distribution.num_categories == 3


# ---- Empirical: known answers of the reference's own tests (reference tests/test_distributions.py) ----------------
def test_empirical_known_answers():
    from pyprob_amd.distributions import Empirical
    torch.manual_seed(1)
    dist = Empirical(torch.tensor([1., 2., 3.]), torch.tensor([1., 2., 3.]))           # :30-71
    assert abs(dist.mean - 2.5752103328704834) < 1e-6 and abs(dist.stddev - 0.6514633893966675) < 1e-6
    assert abs(dist.expectation(torch.sin) - 0.3921678960323334) < 1e-6
    assert abs(dist.map(torch.sin).mean - 0.3921678960323334) < 1e-6
    assert dist.min == 1 and dist.max == 3 and float(dist.mode) == 3 and dist.weighted
    un = dist.unweighted()
    assert abs(un.mean - 2) < 1e-12 and abs(un.stddev - 0.816497) < 1e-5 and not un.weighted
    drawn = Empirical([dist.sample() for _ in range(20000)])
    assert abs(drawn.mean - 2.5752) < 0.03 and abs(drawn.stddev - 0.6515) < 0.03
    assert abs(dist.effective_sample_size - 1.0 / np.sum(dist.weights_numpy() ** 2)) < 1e-12
    assert list(Empirical([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12]).thin(4).values_numpy()) == [1, 4, 7, 10]     # :374-384
    d = Empirical([0, 1, 2, 3, 4, 5])                                                                       # :401-415
    assert d[0:3].get_values() == [0, 1, 2] and d[0] == 0 and d[-1] == 5 and list(d) == [0, 1, 2, 3, 4, 5]
    d = Empirical([2, 2, 2, 2, 3, 3, 3, 4, 4])                                                              # :434-448
    assert d.sample(min_index=0, max_index=3) == 2 and d.sample(min_index=4, max_index=6) == 3
    assert d.sample(min_index=7, max_index=8) == 4
    vals = torch.distributions.Normal(2.0, 5.0).sample((20000,))                                            # :338-353
    r = Empirical(list(vals)).resample(10000)
    assert len(r) == 10000 and abs(r.mean - 2) < 0.25 and abs(r.stddev - 5) < 0.25 and r.metadata['op'] == 'resample'
    z = Empirical(list(torch.randn(20000)))                                                                 # :774-785
    assert abs(z.skewness) < 0.1 and abs(z.kurtosis - 3.0) < 0.15
    e = Empirical(list(torch.distributions.Exponential(1.5).sample((20000,))))                              # :801-813
    assert abs(e.mean - 0.666667) < 0.03 and abs(e.median - 0.462098) < 0.03
    # weighted resampling reproduces the weighted moments; condition keeps the weights of what it keeps (:885-905)
    w = Empirical(list(range(10)), log_weights=list(np.linspace(-3, 0, 10)))
    assert abs(w.resample(40000).mean - w.mean) < 0.05
    c = w.condition(lambda x: x >= 5)
    assert len(c) == 5 and abs(c.mean - np.sum(w.weights_numpy()[5:] * np.arange(5, 10)) / w.weights_numpy()[5:].sum()) < 1e-12
    # concatenation (ParallelModel's merge, model.py:395-404; reference test :674-702)
    a, b = Empirical([1., 2.], log_weights=[0., 1.]), Empirical([3.], log_weights=[2.])
    cat = Empirical(concat_empiricals=[a, b])
    ref = Empirical([1., 2., 3.], log_weights=[0., 1., 2.])
    assert len(cat) == 3 and abs(cat.mean - ref.mean) < 1e-12 and abs(cat.effective_sample_size - ref.effective_sample_size) < 1e-12
    assert Empirical([1., 2.], weights=[1., 3.]).mean == 1.75


def test_prior_and_posterior_return_empiricals_of_traces():
    """Model.prior / posterior (pyprob/model.py:97-117): Empiricals of Trace objects; trace['name'] reads a named
    variable (trace.py:192-196); map / condition on them like reference tests/test_distributions.py:885-905."""
    torch.manual_seed(8)
    model = GaussianWithUnknownMean()
    prior = model.prior(300)
    assert len(prior) == 300 and not prior.weighted and 'obs0' in prior[0] and prior.metadata['op'] == 'prior'
    assert prior[0]['obs0'] is None        # TraceMode.PRIOR leaves an unobserved `observe` without a value (state.py:139)
    mu = prior.map(lambda t: float(t.result))
    assert abs(mu.mean - 1.0) < 0.6
    assert len(prior.condition(lambda t: float(t.result) > 1.0)) < 300
    post = model.posterior(2000, InferenceEngine.IMPORTANCE_SAMPLING, observe={'obs0': 8, 'obs1': 9})
    assert post.weighted and abs(post.map(lambda t: float(t.result)).mean - 7.25) < 0.8
    with pytest.raises(RuntimeError):
        prior[0]['nope']
    res = model.prior_results(50)
    assert len(res) == 50 and np.isfinite(res.mean)


def test_learning_rate_schedules_match_the_reference_formula():
    """POLY1 / POLY2 decay over the trace count (pyprob/nn/inference_network.py:357-379); enum or string."""
    import pyprob_amd
    from pyprob_amd.nn import InferenceNetworkLSTM
    net = InferenceNetworkLSTM(observe_embeddings={'obs0': {'dim': 8}})
    net._learning_rate_init, net._learning_rate_end, net._total_train_traces_end = 1e-3, 1e-6, 1e6
    for kind, power in ((pyprob_amd.LearningRateScheduler.POLY1, 1.0), ('POLY2', 2.0)):
        net._learning_rate_scheduler_type = kind
        for traces in (0, 250000, 999999, 2000000):
            want = (1e-3 - 1e-6) * (max(0.0, 1 - traces / 1e6) ** power) + 1e-6
            assert abs(net._learning_rate(traces) - want) < 1e-15
    net._learning_rate_scheduler_type = pyprob_amd.LearningRateScheduler.NONE
    assert net._learning_rate(12345) == 1e-3
    net._learning_rate_scheduler_type = 'COSINE'
    with pytest.raises(ValueError):
        net._learning_rate(1)
