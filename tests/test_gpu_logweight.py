"""The importance-sampling log-weight path on the device, pinned on the reference's records.

The goldens hold what pyprob itself computed for every particle of a posterior run (tests/golden/make_golden.py):
the prior log_prob of every controlled value (`prior_lp`, state.py:211), the proposal log_prob (`prop_lp`, :212), the
sum of the observed-likelihood terms (`obs_lw`, state.py:147-149) and the trace log-weight (`lw`, trace.py:123-125).
Here all of them come out of the C-ABI kernels that the product's posterior runs use - pp_is_step, pp_logweight_terms,
pp_logweight_accumulate, pp_is_stats - and must agree to 1e-4 relative (BASELINE.json north_star)."""
import numpy as np
import pytest

from conftest import load_golden
from helpers import engine_from_golden
from oracle import ic_oracle as O

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')

# likelihood of the two observations of every golden program (tests/golden/make_golden.py): Normal(result, sigma)
LIKELIHOOD_STDDEV = {'gum': 2.0 ** 0.5, 'gumm': 2.0 ** 0.5, 'gumm2': 2.0 ** 0.5, 'gumd': 2.0 ** 0.5, 'ff': 2.0 ** 0.5, 'cat': 0.8, 'poi': 0.8, 'ber': 0.8, 'ffc': 0.8}


class _Dist:
    """The duck type ISRunner.dist_term reads (pyprob/distributions/*.py attribute names)."""

    def __init__(self, name, **kw):
        self.name = name
        self.__dict__.update(kw)


def _prior_dist(dist_name, row):
    if dist_name == 'Normal':
        return _Dist('Normal', mean=row[0], stddev=row[1])
    if dist_name == 'Uniform':
        return _Dist('Uniform', low=row[0], high=row[1])
    if dist_name == 'Poisson':
        return _Dist('Poisson', rate=row[0])
    if dist_name == 'Bernoulli':
        return _Dist('Bernoulli', probs=row[0])
    if dist_name == 'Categorical':
        return _Dist('Categorical', probs=np.asarray(row, np.float32), num_categories=len(row))
    raise RuntimeError(dist_name)


def _head_prior(info, row, dev):
    pr = row[:2] if info.dist_name != 'Poisson' else np.array([0.0, 40.0], np.float32)
    return torch.tensor(np.asarray(pr, np.float32).reshape(1, 2), device=dev)


def test_device_log_weights_match_reference_records(golden):
    """Every particle of the recorded posterior run, scored statement by statement like state.sample / state.observe /
    Trace.end do - on the device: lw += log p(v) - log q(v) per controlled variable, lw += log p(y_j | result) per
    observation. Compared with the reference's own prior_lp, prop_lp, obs_lw and lw."""
    from pyprob_amd.is_engine import ISRunner
    case, meta, params, batch, loss, isr = golden
    eng = engine_from_golden(meta, params)
    run = ISRunner(eng)
    run.init(isr['observe'])
    dev = eng.device
    addresses = meta['is_addresses']
    n_traces = len(isr['trace_len'])
    off = np.concatenate([[0], np.cumsum(isr['trace_len'])])
    lw = torch.zeros(n_traces, dtype=torch.float32, device=dev)          # the accumulator under test
    lik = torch.zeros(n_traces, dtype=torch.float32, device=dev)
    prior_lp = torch.zeros(len(isr['value']), dtype=torch.float32, device=dev)
    logq_all = torch.zeros(len(isr['value']), dtype=torch.float32, device=dev)
    sigma = torch.tensor([LIKELIHOOD_STDDEV[case]], dtype=torch.float32, device=dev)
    obs = [torch.tensor([float(y)], dtype=torch.float32, device=dev) for y in isr['observe']]
    for b in range(n_traces):
        run.begin(1)
        prev = None
        for t in range(int(isr['trace_len'][b])):
            r = int(off[b] + t)
            a = eng.spec.address_id[addresses[isr['addr'][r]]]
            info = eng.spec.addresses[a]
            v = torch.tensor([isr['value'][r]], dtype=torch.float32, device=dev)
            _, logq = run.step(a, prev, _head_prior(info, isr['prior'][r], dev), value_in=v)
            ncat = info.num_categories or 0
            term = run.dist_term(_prior_dist(info.dist_name, isr['prior'][r, :ncat] if ncat else isr['prior'][r]))
            assert term is not None, info.dist_name
            # state.py:211-217 in one pass: + log p(v) - log q(v)
            run.accumulate_terms(lw[b:b + 1], [(term, v, 1.0), (2, None, None, logq, -1.0)])
            prior_lp[r:r + 1] = run.log_prob(term, v)
            logq_all[r:r + 1] = logq
            prev = a
        # state.observe for obs0, obs1 (state.py:147-149): Normal(result, sigma) likelihoods of the fixed observations
        mu = torch.tensor([isr['result'][b]], dtype=torch.float32, device=dev)
        terms = [(0, mu, sigma, y, 1.0) for y in obs]
        run.accumulate_terms(lw[b:b + 1], terms)
        run.accumulate_terms(lik[b:b + 1], terms, overwrite=True)
    np.testing.assert_allclose(logq_all.cpu().numpy(), isr['prop_lp'], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(prior_lp.cpu().numpy(), isr['prior_lp'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(lik.cpu().numpy(), isr['obs_lw'], rtol=1e-4, atol=1e-5)
    got = lw.cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(got, isr['lw'], rtol=1e-4, atol=1e-4)
    # the same weights through the float64 oracle (re-scoring of the records) agree as well
    if meta.get('network', 'lstm') == 'lstm':
        net = O.Net(params, meta['obs_names'], K=meta['mixture_components'])
        dist_names = [eng.spec.addresses[eng.spec.address_id[a]].dist_name for a in addresses]
        _, _, _, lw_ref = O.is_rescore(net, isr['observe'], isr['trace_len'], isr['addr'], isr['value'], isr['prior'],
                                       addresses, dist_names)
        np.testing.assert_allclose(got, lw_ref + isr['obs_lw'], rtol=1e-4, atol=1e-4)


def test_batched_log_weights_equal_per_particle(golden):
    """The product runs the statements for all particles of a path at once (n-wide tensors, per-particle parameters).
    Traces of the record that share an address sequence are scored as ONE lock-step group; the device log-weights must
    equal the reference's per-trace numbers."""
    case, meta, params, batch, loss, isr = golden
    _score_in_groups(case, engine_from_golden(meta, params), isr, meta['is_addresses'])


def _score_in_groups(case, eng, isr, addresses):
    from pyprob_amd.is_engine import ISRunner
    run = ISRunner(eng)
    run.init(isr['observe'])
    dev = eng.device
    off = np.concatenate([[0], np.cumsum(isr['trace_len'])])
    groups = {}
    for b in range(len(isr['trace_len'])):
        groups.setdefault(tuple(isr['addr'][off[b]:off[b + 1]].tolist()), []).append(b)
    sigma = torch.tensor([LIKELIHOOD_STDDEV[case]], dtype=torch.float32, device=dev)
    for seq, members in groups.items():
        n = len(members)
        members = np.asarray(members)
        run.begin(n)
        lw = torch.empty(n, dtype=torch.float32, device=dev)
        prev = None
        for t, ai in enumerate(seq):
            rows = off[members] + t
            a = eng.spec.address_id[addresses[ai]]
            info = eng.spec.addresses[a]
            v = torch.tensor(isr['value'][rows], dtype=torch.float32, device=dev)
            pr = isr['prior'][rows]
            head = pr[:, :2].copy() if info.dist_name != 'Poisson' else np.tile(np.array([[0.0, 40.0]], np.float32), (n, 1))
            _, logq = run.step(a, prev, torch.tensor(head, dtype=torch.float32, device=dev).contiguous(), value_in=v)
            if info.dist_name == 'Categorical':
                C = info.num_categories
                d = _Dist('Categorical', probs=pr[:, :C].copy(), num_categories=C)       # per-particle probability rows
            elif info.dist_name == 'Normal':
                d = _Dist('Normal', mean=pr[:, 0].copy(), stddev=pr[:, 1].copy())
            elif info.dist_name == 'Uniform':
                d = _Dist('Uniform', low=pr[:, 0].copy(), high=pr[:, 1].copy())
            elif info.dist_name == 'Poisson':
                d = _Dist('Poisson', rate=pr[:, 0].copy())
            else:
                d = _Dist('Bernoulli', probs=pr[:, 0].copy())
            run.accumulate_terms(lw, [(run.dist_term(d), v, 1.0), (2, None, None, logq, -1.0)], overwrite=(t == 0))
            prev = a
        mu = torch.tensor(isr['result'][members], dtype=torch.float32, device=dev)
        run.accumulate_terms(lw, [(0, mu, sigma, torch.tensor([float(y)], device=dev), 1.0) for y in isr['observe']])
        np.testing.assert_allclose(lw.cpu().numpy(), isr['lw'][members], rtol=1e-4, atol=1e-4)


def test_importance_statistics_match_float64_oracle():
    """pp_is_stats (Empirical.finalize / expectation / effective_sample_size, empirical.py:298-309, 451-466, 758-766)
    against the float64 definitions on the same log-weights: ESS = 1 / sum softmax(lw)^2 (util.py:398-399), weighted
    mean and variance."""
    from pyprob_amd.engine import ICEngine   # noqa: F401  (loads the library)
    from pyprob_amd.is_engine import ISRunner
    meta, params, batch, loss, isr = load_golden('gum')
    run = ISRunner(engine_from_golden(meta, params))
    rng = np.random.default_rng(3)
    for n in (1, 7, 4096, 300001):
        lw = (rng.standard_normal(n) * 3.0 - 40.0).astype(np.float32)
        x = rng.standard_normal(n).astype(np.float32) + 7.0
        if n > 16:
            lw[5] = -np.inf       # dropped like Model._traces drops them (model.py:64-66)
            lw[11] = np.nan
        st = run.stats(torch.tensor(lw, device=run.dev), torch.tensor(x, device=run.dev))
        ok = np.isfinite(lw)
        l64, x64 = lw[ok].astype(np.float64), x[ok].astype(np.float64)
        ess = O.effective_sample_size(l64)
        w = np.exp(l64 - O.logsumexp(l64, axis=0))
        mean = float((w * x64).sum())
        var = float((w * x64 * x64).sum() - mean * mean)
        assert st['count'] == ok.sum()
        assert abs(st['ess'] - ess) <= 1e-9 * ess, (n, st['ess'], ess)
        assert abs(st['mean'] - mean) <= 1e-9 * abs(mean), (n, st['mean'], mean)
        assert abs(st['var'] - var) <= 1e-7 * max(var, 1e-12) + 1e-12, (n, st['var'], var)
        assert abs(st['max_lw'] - l64.max()) == 0.0


def test_prior_log_prob_kernels_against_oracle():
    """Every family of pp_logweight_accumulate on random arguments vs the oracle's float64 formulas (which are pinned on
    the reference's known-answer tests, tests/test_oracle.py)."""
    from pyprob_amd.is_engine import ISRunner
    meta, params, batch, loss, isr = load_golden('gum')
    run = ISRunner(engine_from_golden(meta, params))
    dev = run.dev
    rng = np.random.default_rng(5)
    n = 5000

    def dev_t(a):
        return torch.tensor(np.asarray(a, np.float32), device=dev)
    v = rng.standard_normal(n).astype(np.float32) * 2
    mu, sd = rng.standard_normal(n).astype(np.float32), rng.uniform(0.1, 3, n).astype(np.float32)
    got = run.log_prob(run.dist_term(_Dist('Normal', mean=mu, stddev=sd)), dev_t(v)).cpu().numpy()
    np.testing.assert_allclose(got, O.normal_log_prob(v.astype(np.float64), mu, sd), rtol=1e-5, atol=1e-5)
    lo, hi = -rng.uniform(0.5, 2, n).astype(np.float32), rng.uniform(0.5, 2, n).astype(np.float32)
    got = run.log_prob(run.dist_term(_Dist('Uniform', low=lo, high=hi)), dev_t(v)).cpu().numpy()
    ref = O.uniform_log_prob(v.astype(np.float64), lo, hi)
    assert np.array_equal(np.isneginf(got), np.isneginf(ref))
    np.testing.assert_allclose(got[np.isfinite(ref)], ref[np.isfinite(ref)], rtol=1e-5, atol=1e-6)
    rate = rng.uniform(0.2, 9, n).astype(np.float32)
    k = np.concatenate([rng.poisson(4.0, n // 2), rng.uniform(0, 12, n - n // 2)]).astype(np.float32)   # counts and reals
    got = run.log_prob(run.dist_term(_Dist('Poisson', rate=rate)), dev_t(k)).cpu().numpy()
    np.testing.assert_allclose(got, O.poisson_log_prob(k.astype(np.float64), rate), rtol=2e-5, atol=2e-5)
    p = rng.uniform(0, 1, n).astype(np.float32)
    p[:3] = [0.0, 1.0, 0.5]
    bv = (rng.uniform(0, 1, n) < 0.5).astype(np.float32)
    got = run.log_prob(run.dist_term(_Dist('Bernoulli', probs=p)), dev_t(bv)).cpu().numpy()
    np.testing.assert_allclose(got, O.bernoulli_log_prob(bv.astype(np.float64), p), rtol=1e-5, atol=1e-5)
    C = 6
    probs = rng.uniform(0.01, 1, (n, C)).astype(np.float32)
    idx = rng.integers(0, C, n).astype(np.float32)
    got = run.log_prob(run.dist_term(_Dist('Categorical', probs=probs, num_categories=C)), dev_t(idx)).cpu().numpy()
    ref = np.array([O.categorical_log_prob(idx[i:i + 1], probs[i].astype(np.float64))[0] for i in range(n)])
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5)
    shared = run.log_prob(run.dist_term(_Dist('Categorical', probs=probs[0], num_categories=C)), dev_t(idx)).cpu().numpy()
    np.testing.assert_allclose(shared, O.categorical_log_prob(idx, probs[0].astype(np.float64)), rtol=1e-5, atol=1e-5)


# ---- executors on the device: what they hand out, re-scored by the oracle ------------------------------------------
def test_lock_step_gumm_run_rescored_by_the_oracle():
    """A lock-step run of the Marsaglia program (stochastic control flow: one execution per path, replayed prefixes,
    masked accumulation, gathered LSTM rows) on the device: (path, per-statement values, final log-weight) of every
    particle re-scored with the oracle's batch-1 `_infer_step` restatement (state.py:203-219, trace.py:123-125)."""
    import math
    import warnings
    from is_helpers import lockstep_network, rescore_lockstep_run
    from pyprob_amd.state import InferenceEngine
    model, net, meta, params = lockstep_network('cuda:0')
    observe = {'obs0': 8, 'obs1': 9}
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        post = model.posterior_results(3000, InferenceEngine.IMPORTANCE_SAMPLING_WITH_INFERENCE_NETWORK, observe=observe,
                                       lock_step=True, seed=2)
    assert post.num_paths > 3
    post.statement_log = [{a: (v.cpu(), i) for a, (v, i) in entry.items()} for entry in post.statement_log]
    lw_ref, results = rescore_lockstep_run(post, net, meta, params, observe, math.sqrt(2))
    got = post._all_log_weights.cpu().numpy()
    ok = np.isfinite(lw_ref)
    assert ok.sum() > 0.99 * len(ok)
    np.testing.assert_allclose(got[ok], lw_ref[ok], rtol=1e-4, atol=2e-4)
    np.testing.assert_allclose(post._all_values.cpu().numpy()[ok], results[ok], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('case', ['gum', 'gumm', 'cat', 'poi', 'ber', 'ff', 'gumm2'])
def test_coroutine_run_rescored_by_the_oracle(case):
    """The reference's programs AS WRITTEN (`while float(s) >= 1`) through the particle-coroutine scheduler on the
    device: every particle's trace (addresses, values, priors) re-scored by the oracle equals its device log-weight."""
    import math
    import warnings
    pytest.importorskip('greenlet')
    from is_helpers import network_from_golden, rescore
    from test_coroutine import CASES
    _, program, observe, sigma = [c for c in CASES if c[0] == case][0]
    net, meta, params, isr = network_from_golden(case, 'cuda:0')
    model = program()
    model._inference_network = net
    n = 300
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        post = model._traces_coroutines(n, observe, map_func=lambda t: t, seed=5)
    traces = post.get_values()
    lw = np.array([t.log_importance_weight for t in traces])
    ref = rescore(case, meta, params, traces, observe, sigma)
    ok = np.isfinite(ref)
    assert len(traces) + (n - post.length) == n and ok.all()
    np.testing.assert_allclose(lw, ref, rtol=1e-4, atol=2e-4)
    st = post.coroutine_stats
    assert st['statements'] == sum(len(t.variables_controlled) for t in traces)


def _identity(trace):
    return trace


def test_sharded_coroutine_run_rescored_by_the_oracle():
    """Particle shards in forked worker processes, the parent serving the device (pyprob_amd/coroutine.py
    ShardedCoroutineIS): traces come back from the workers, their device log-weights equal the oracle's re-scoring."""
    import math
    import warnings
    pytest.importorskip('greenlet')
    from is_helpers import network_from_golden, rescore
    from models import GaussianWithUnknownMeanMarsaglia
    net, meta, params, isr = network_from_golden('gumm', 'cuda:0')
    model = GaussianWithUnknownMeanMarsaglia()
    model._inference_network = net
    observe = {'obs0': 8, 'obs1': 9}
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        post = model._traces_coroutines(600, observe, map_func=_identity, seed=3, num_workers=4)
    traces = post.get_values()
    assert post.coroutine_stats['workers'] == 4 and len(traces) + (600 - post.length) == 600
    lw = np.asarray(post.log_weights_numpy(), np.float64)
    ref = rescore('gumm', meta, params, traces, observe, math.sqrt(2))
    np.testing.assert_allclose(lw, ref, rtol=1e-4, atol=2e-4)
