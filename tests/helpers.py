"""Shared helpers of the parity tests: build an ICEngine / PackedBatch from the golden vectors."""
import numpy as np

from pyprob_amd.spec import NetSpec


def spec_from_golden(meta, params):
    depths = meta.get('observe_embedding_depths', {})
    obs = {n: {'dim': meta['observe_embedding_dims'][n], 'input_dim': 1, 'depth': depths.get(n, 2)} for n in meta['obs_names']}
    spec = NetSpec(obs, lstm_dim=meta['lstm_dim'], proposal_mixture_components=meta['mixture_components'],
                   network=meta.get('network', 'lstm'), lstm_depth=meta.get('lstm_depth', 1))
    pairs = list(zip(meta['addresses'], meta['dist_names']))
    # addresses the network knows but this batch does not contain (GUMM): dist type from the address suffix
    for k in params:
        if k.startswith('_layers_proposal.') and k.endswith('._ff._layers.0.weight'):
            a = k[len('_layers_proposal.'):-len('._ff._layers.0.weight')]
            if a not in meta['addresses']:
                suffix = a.split('__')[-2]
                pairs.append((a, [d for d in ('Normal', 'Uniform', 'Categorical', 'Poisson', 'Bernoulli') if suffix.startswith(d)][0]))
    for a, d in pairs:
        ncat = None
        if d == 'Categorical':
            ncat = params['_layers_proposal.%s._ff._layers.1.weight' % a].shape[0]
        spec.add_address(a, d, ncat)
    return spec


def engine_from_golden(meta, params, device='cuda:0'):
    from pyprob_amd.engine import ICEngine
    spec = spec_from_golden(meta, params)
    eng = ICEngine(spec, device=device, seed=0)
    assert set(spec.tensors.keys()) == set(params.keys()), set(spec.tensors.keys()) ^ set(params.keys())
    for n in spec.tensors:
        assert spec.tensors[n][1] == params[n].shape, (n, spec.tensors[n][1], params[n].shape)
    eng.load_state_dict(params)
    return eng


def packed_from_golden(meta, batch, spec):
    from pyprob_amd.packed import PackedBatch
    from pyprob_amd.packed import bernoulli_group_stats
    ids = np.array([spec.address_id[meta['addresses'][i]] for i in batch['addr_idx']], np.int64)
    prior = head_prior(meta, batch['addr_idx'], batch['prior'])
    bernoulli = [a for a, info in enumerate(spec.addresses) if info.dist_name == 'Bernoulli']
    if bernoulli:
        prior = bernoulli_group_stats(batch['trace_len'], ids, batch['values'], prior, bernoulli)
    return PackedBatch.from_ragged(batch['trace_len'], ids, batch['values'], prior, batch['obs'], len(spec.addresses))


def head_prior(meta, addr_idx, prior, dist_names=None):
    """Prior parameters as the proposal heads read them: the golden files hold the prior's own parameters (Poisson:
    the rate), the Poisson head works on the fixed interval [0, 40] (pyprob_amd.packed.distribution_params)."""
    from pyprob_amd.packed import POISSON_LOW_HIGH
    names = np.asarray(meta['dist_names'] if dist_names is None else dist_names)[np.asarray(addr_idx)]
    out = np.zeros((len(prior), 2), np.float32)
    w = min(2, prior.shape[1])
    out[:, :w] = prior[:, :w]
    out[names == 'Poisson'] = POISSON_LOW_HIGH
    return out


def rel_err(got, ref):
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    return float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-12))


def grad_check(label, got, ref, bar, abs_floor=0.0):
    """Gradient tensor against the float64 oracle: max |got - ref| < bar * max |ref| + abs_floor. The bars of the callers are
    <= 10x the error MEASURED on MI355X (profiles/r04_grad_errors.jsonl, recorded with PP_TEST_RECORD_ERRORS=<file>: every
    check appends its error there), not a generic tolerance: a dropped small contribution shows up. abs_floor: the ragged
    cases run on freshly initialised networks whose gradients are ~1e-6 - their fp32 summation error (measured 5e-9, the
    same absolute size as in the cases with gradients of 1e-2) is not small RELATIVE to such a tensor."""
    import json
    import os
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    abs_err = float(np.abs(got - ref).max())
    ref_max = float(np.abs(ref).max())
    path = os.environ.get('PP_TEST_RECORD_ERRORS')
    if path:
        with open(path, 'a') as f:
            f.write(json.dumps(dict(check=label, err=abs_err / max(ref_max, 1e-30), abs_err=abs_err, bar=bar, abs_floor=abs_floor,
                                    ref_max=ref_max)) + '\n')
        return abs_err
    assert abs_err < bar * ref_max + abs_floor, (label, abs_err, ref_max, bar, abs_floor)
    return abs_err


def synthetic_gum_arrays(n, seed=0):
    """GaussianUnknownMean prior traces in trace-major arrays: mu ~ N(1, sqrt5), y0,y1 ~ N(mu, sqrt2)
    (the model of the reference's tests/test_inference.py:97-109)."""
    rng = np.random.default_rng(seed)
    mu = (1.0 + np.sqrt(5.0) * rng.standard_normal(n)).astype(np.float32)
    obs = (mu[:, None] + np.sqrt(2.0) * rng.standard_normal((n, 2))).astype(np.float32)
    prior = np.tile(np.array([[1.0, np.sqrt(5.0)]], np.float32), (n, 1))
    return dict(trace_len=np.ones(n, np.int32), addr_idx=np.zeros(n, np.int32), values=mu, prior=prior, obs=obs)


def synthetic_gumm_arrays(n, seed=0, max_iter=6):
    """GaussianUnknownMeanMarsaglia traces (tests/test_inference.py:252-275): pairs x,y ~ U(-1,1) until
    x^2+y^2 < 1; addresses alternate x_k, y_k with k the loop iteration. Returns arrays + address list."""
    rng = np.random.default_rng(seed)
    trace_len, addr_idx, values, obs = [], [], [], []
    for _ in range(n):
        k = 0
        while True:
            x, y = rng.uniform(-1, 1, 2)
            addr_idx += [2 * k, 2 * k + 1]
            values += [x, y]
            k += 1
            s = x * x + y * y
            if s < 1 or k >= max_iter:
                break
        if s >= 1:
            s = 0.5
        trace_len.append(2 * k)
        mu = 1.0 + np.sqrt(5.0) * x * np.sqrt(-2 * np.log(s) / s)
        obs.append(mu + np.sqrt(2.0) * rng.standard_normal(2))
    R = len(values)
    prior = np.tile(np.array([[-1.0, 1.0]], np.float32), (R, 1))
    n_addr = 2 * max_iter
    addresses = ['a%d__%s__Uniform__%d' % (i, 'xy'[i % 2], i // 2 + 1) for i in range(n_addr)]
    return dict(trace_len=np.array(trace_len, np.int32), addr_idx=np.array(addr_idx, np.int32),
                values=np.array(values, np.float32), prior=prior, obs=np.array(obs, np.float32)), addresses
