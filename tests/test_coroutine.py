"""The particle-coroutine scheduler (pyprob_amd/coroutine.py) on CPU: the HOST logic - parking in `sample`, grouping by
(address, previous address), gathering / scattering the particles' LSTM state, the deferred likelihood terms, the
per-particle context switch of the trace runtime - executed with the oracle-backed CPU kernels of the operators
(tests/oracle_ops.py), on the reference's own program source (`while float(s) >= 1`, tests/test_inference.py:252-275).
Every particle's recorded trace is re-scored with the oracle's batch-1 `_infer_step` restatement: the scheduler must
reproduce log w = sum (log p - log q) + sum log p(y | .) for exactly the values it handed out."""
import math
import warnings

import numpy as np
import pytest

import oracle_ops  # noqa: F401  registers the CPU kernels of pyprob_hip::*
from is_helpers import lockstep_network, network_from_golden, rescore, rescore_lockstep_run
from models import (BernoulliThenNormal, CategoricalThenNormal, GaussianWithUnknownMean, GaussianWithUnknownMeanMarsaglia,
                    PoissonThenNormal)
from oracle import ic_oracle as O

pytest.importorskip('greenlet')


CASES = [('gum', GaussianWithUnknownMean, {'obs0': 8, 'obs1': 9}, math.sqrt(2)),
         ('gumm', GaussianWithUnknownMeanMarsaglia, {'obs0': 8, 'obs1': 9}, math.sqrt(2)),
         ('cat', CategoricalThenNormal, {'obs0': 1.2, 'obs1': 0.7}, 0.8),
         ('poi', PoissonThenNormal, {'obs0': 2.2, 'obs1': 1.7}, 0.8),
         ('ber', BernoulliThenNormal, {'obs0': 1.2, 'obs1': 0.7}, 0.8),
         ('ff', GaussianWithUnknownMeanMarsaglia, {'obs0': 8, 'obs1': 9}, math.sqrt(2)),
         ('gumm2', GaussianWithUnknownMeanMarsaglia, {'obs0': 8, 'obs1': 9}, math.sqrt(2))]        # nn.LSTM depth 2


@pytest.mark.parametrize('case,program,observe,sigma', CASES, ids=[c[0] for c in CASES])
def test_coroutine_log_weights_equal_rescored_traces(case, program, observe, sigma):
    net, meta, params, isr = network_from_golden(case)
    model = program()
    model._inference_network = net
    n = 40
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        post = model._traces_coroutines(n, observe, map_func=lambda t: t, seed=5)
    traces = post.get_values()
    assert len(traces) == n
    for tr in traces:     # every controlled address was served by the network (same strings as the reference's)
        assert all(v.address in net._engine.spec.address_id for v in tr.variables_controlled)
    lw = np.array([t.log_importance_weight for t in traces])
    ref = rescore(case, meta, params, traces, observe, sigma)
    np.testing.assert_allclose(lw, ref, rtol=2e-5, atol=2e-5)
    st = post.coroutine_stats
    assert st['statements'] == sum(len(t.variables_controlled) for t in traces)
    if case in ('gumm', 'ff'):      # variable-length traces: several rounds, fewer particles each
        lens = [len(t.variables_controlled) for t in traces]
        assert st['rounds'] == max(lens) and min(lens) == 2


def test_coroutine_results_match_lock_step_free_statistics():
    """posterior_results(lock_step=False) = coroutines; GUM with the golden network: weighted mean of a
    self-normalised estimate is finite, ESS > 1, results are a tensor (device statistics available)."""
    net, meta, params, isr = network_from_golden('gum')
    model = GaussianWithUnknownMean()
    model._inference_network = net
    from pyprob_amd.state import InferenceEngine
    post = model.posterior_results(200, InferenceEngine.IMPORTANCE_SAMPLING_WITH_INFERENCE_NETWORK,
                                   observe={'obs0': 8, 'obs1': 9}, lock_step=False, seed=3)
    assert post.length == 200 and np.isfinite(post.mean) and post.effective_sample_size > 1
    assert abs(post.device_stats['ess'] - post.effective_sample_size) < 1e-6 * post.effective_sample_size


def test_unknown_address_falls_back_to_the_prior():
    """A program with an address the network has no layers for: the prior is the proposal (log p - log q = 0), the
    statement is not served by the device, and the NEXT statement's 'previous variable' is unknown too."""
    net, meta, params, isr = network_from_golden('gum')

    class Extra(GaussianWithUnknownMean):
        def forward(self):
            import pyprob_amd as pyprob
            from pyprob_amd.distributions import Normal
            z = pyprob.sample(Normal(0., 1.))          # unknown to the golden network
            mu = pyprob.sample(Normal(self.prior_mean, self.prior_stddev))   # different call site than training: unknown too
            pyprob.observe(Normal(mu + 0 * z, self.likelihood_stddev), name='obs0')
            pyprob.observe(Normal(mu, self.likelihood_stddev), name='obs1')
            return mu
    model = Extra()
    model._inference_network = net
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        post = model._traces_coroutines(6, {'obs0': 8.0, 'obs1': 9.0}, map_func=lambda t: t)
    assert any('Using prior' in str(x.message) for x in w)
    for tr in post.get_values():
        mu = float(tr.result)
        ref = float(O.normal_log_prob(8.0, mu, math.sqrt(2))) + float(O.normal_log_prob(9.0, mu, math.sqrt(2)))
        assert abs(tr.log_importance_weight - ref) < 1e-5


def test_lock_step_run_rescored_by_the_oracle():
    """The lock-step path executor (state.LockStepState: one execution per control-flow path, replayed prefixes, masked
    accumulation) on a program with stochastic control flow: every particle's log-weight equals the oracle's re-scoring of
    the values that particle drew."""
    from pyprob_amd.state import InferenceEngine
    model, net, meta, params = lockstep_network()
    observe = {'obs0': 8, 'obs1': 9}
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        post = model.posterior_results(48, InferenceEngine.IMPORTANCE_SAMPLING_WITH_INFERENCE_NETWORK, observe=observe,
                                       lock_step=True, seed=2)
    assert post.num_paths > 1
    lw_ref, results = rescore_lockstep_run(post, net, meta, params, observe, math.sqrt(2))
    np.testing.assert_allclose(post._all_log_weights.numpy(), lw_ref, rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(post._all_values.numpy(), results, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('case,program,observe,sigma', [CASES[1], CASES[2]], ids=['gumm', 'cat'])
def test_sharded_coroutines_equal_rescored_traces(case, program, observe, sigma):
    """The particles spread over forked worker processes (the parent serves the operators): the workers' parked
    statements are merged by (address, previous address) each round; every returned trace re-scores to its log-weight."""
    net, meta, params, isr = network_from_golden(case)
    model = program()
    model._inference_network = net
    n = 45
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        post = model._traces_coroutines(n, observe, map_func=_identity, seed=5, num_workers=3)
    traces = post.get_values()
    assert len(traces) == n and post.coroutine_stats['workers'] == 3
    lw = post.log_weights_numpy() if hasattr(post, 'log_weights_numpy') else np.asarray(post.log_weights)
    ref = rescore(case, meta, params, traces, observe, sigma)
    np.testing.assert_allclose(np.asarray(lw, np.float64), ref, rtol=2e-5, atol=2e-5)
    st = post.coroutine_stats
    assert st['statements'] == sum(len(t.variables_controlled) for t in traces)
    assert st['group_calls'] < st['statements']
    # plain results come back as one tensor with device statistics
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        p2 = model._traces_coroutines(40, observe, seed=6, num_workers=2)
    assert p2.length == 40 and np.isfinite(p2.mean) and 'ess' in p2.device_stats


def test_persistent_workers_are_reused_and_give_the_same_particles(monkeypatch):
    """The worker pool is forked once per (program, worker count): a second posterior call runs in the same processes and,
    with the same seed, returns the same particles as a call with per-call forks (PP_IS_POOL=0)."""
    from pyprob_amd import coroutine as CO
    case, program, observe, sigma = CASES[1]
    net, meta, params, isr = network_from_golden(case)
    model = program()
    model._inference_network = net
    CO.close_worker_pools()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        a = model._traces_coroutines(30, observe, seed=11, num_workers=2)
        pids = [pr.pid for pool in CO._POOLS.values() for pr in pool.procs]
        assert len(CO._POOLS) == 1 and len(pids) == 2
        b = model._traces_coroutines(30, observe, seed=11, num_workers=2)
        assert [pr.pid for pool in CO._POOLS.values() for pr in pool.procs] == pids
        monkeypatch.setenv('PP_IS_POOL', '0')
        c = model._traces_coroutines(30, observe, seed=11, num_workers=2)
    np.testing.assert_allclose(np.asarray(a.log_weights), np.asarray(b.log_weights), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(np.asarray(a.log_weights), np.asarray(c.log_weights), rtol=1e-6, atol=1e-6)
    CO.close_worker_pools()
    assert not CO._POOLS


def test_pooled_workers_score_the_observations_of_the_current_call(monkeypatch):
    """Persistent workers were forked during an EARLIER posterior call: the observed values, trace mode, engine and
    likelihood_importance of the current call travel with every job (coroutine._runtime_snapshot). A pooled call with
    obs2 after a pooled call with obs1 must equal a fresh-fork call with obs2 - and every trace re-scores with obs2."""
    from pyprob_amd import coroutine as CO
    case, program, obs1, sigma = CASES[1]
    obs2 = {'obs0': 2.5, 'obs1': -1.0}
    net, meta, params, isr = network_from_golden(case)
    model = program()
    model._inference_network = net
    CO.close_worker_pools()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model._traces_coroutines(30, obs1, seed=3, num_workers=2)
        pids = [pr.pid for pool in CO._POOLS.values() for pr in pool.procs]
        pooled = model._traces_coroutines(30, obs2, map_func=_identity, seed=4, num_workers=2)
        assert [pr.pid for pool in CO._POOLS.values() for pr in pool.procs] == pids       # same processes
        monkeypatch.setenv('PP_IS_POOL', '0')
        fresh = model._traces_coroutines(30, obs2, map_func=_identity, seed=4, num_workers=2)
    lw_pooled, lw_fresh = np.asarray(pooled.log_weights, np.float64), np.asarray(fresh.log_weights, np.float64)
    np.testing.assert_allclose(lw_pooled, lw_fresh, rtol=1e-6, atol=1e-6)
    ref = rescore(case, meta, params, pooled.get_values(), obs2, sigma)
    np.testing.assert_allclose(lw_pooled, ref, rtol=2e-5, atol=2e-5)
    monkeypatch.delenv('PP_IS_POOL')
    # a hyper-parameter of the model changed between two calls reaches the workers too
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model.likelihood_stddev = 0.9
        pooled = model._traces_coroutines(30, obs2, map_func=_identity, seed=5, num_workers=2)
    ref = rescore(case, meta, params, pooled.get_values(), obs2, 0.9)
    np.testing.assert_allclose(np.asarray(pooled.log_weights, np.float64), ref, rtol=2e-5, atol=2e-5)
    CO.close_worker_pools()


def test_a_pool_is_not_reused_for_another_model_object():
    from pyprob_amd import coroutine as CO
    case, program, observe, sigma = CASES[1]
    net, meta, params, isr = network_from_golden(case)
    model = program()
    model._inference_network = net
    CO.close_worker_pools()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model._traces_coroutines(20, observe, seed=1, num_workers=2)
    (key, pool), = CO._POOLS.items()
    other = program()
    assert pool.owns(model.forward) and not pool.owns(other.forward)
    CO.close_worker_pools()


def _identity(trace):
    return trace


@pytest.mark.parametrize('case', ['gum', 'gumm'])
def test_ten_thousand_reference_particles_through_the_runner_on_host_buffers(case):
    """The scoring loop of tests/test_gpu_ten_thousand_particles.py (ISRunner.begin / step(value_in=...) / dist_term /
    accumulate_terms in lock-step groups) with the engine's buffers on the host and the operators backed by the oracle: the
    10^4 particles the reference sampled and scored (tests/golden/make_is_10k.py) come out at 1e-4."""
    import os
    from conftest import GOLDEN, load_golden
    from helpers import spec_from_golden
    from test_gpu_logweight import _score_in_groups
    meta, params, batch, loss, isr = load_golden(case)
    eng = oracle_ops.CpuBufferEngine(spec_from_golden(meta, params))
    eng._use_ops = True
    eng.load_state_dict(params)
    big = dict(np.load(os.path.join(GOLDEN, case + '_is10k.npz')))
    _score_in_groups(case, eng, big, [str(a) for a in big['addresses']])
