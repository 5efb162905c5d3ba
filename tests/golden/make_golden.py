#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE (pyprob v1.5.0) on CPU.

Runs only in the build container (needs /root/reference, which does not exist on the GPU box):

    python tests/golden/make_golden.py

The reference is imported read-only with the import stubs in oracle/refstubs (termcolor, sqlitedict,
zmq, flatbuffers, pydotplus are not installed here; none of them is on the hot path). Nothing from the
reference is copied: we call its public API and record inputs/outputs:

  * <case>_net.npz        state_dict of a briefly trained InferenceNetworkLSTM (names in <case>_meta.json)
  * <case>_batch.npz      one minibatch of traces in plain arrays (values, prior params, observations, addresses)
  * <case>_loss.npz       reference `_loss(batch)` (pyprob/nn/inference_network_lstm.py:136-220), per-sub-batch
                          LSTM input/output, per-row proposal log_prob, and every parameter gradient of loss.backward()
  * <case>_is.npz         importance-sampling records from posterior(IMPORTANCE_SAMPLING_WITH_INFERENCE_NETWORK)
                          (pyprob/state.py:203-219, pyprob/trace.py:123-125): sampled values, prior log_prob,
                          proposal parameters, proposal log_prob, per-trace log_importance_weight.

Case gumd: the gum program with observe embeddings of depth 3 (obs0) and 1 (obs1).
Case gumm2: the gumm program with lstm_depth=2 (stacked nn.LSTM layers, lstm_dim=32).
Cases ff / ffc: the same records for InferenceNetworkFeedForward (pyprob/nn/inference_network_feedforward.py) on the
gumm / cat programs (no LSTM records). Cases: gum (GaussianUnknownMean, tests/test_inference.py:97-109), gumm (…Marsaglia, :252-275), both with
lstm_dim=64 so the fixtures stay small, and cat (a Categorical->Normal toy model exercising
ProposalCategoricalCategorical and the one-hot sample embedding).
"""
import json
import math
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(REPO, 'oracle', 'refstubs'))
sys.path.insert(1, '/root/reference')

import numpy as np  # noqa: E402
import torch  # noqa: E402
import pyprob  # noqa: E402
from pyprob import Model, InferenceEngine, InferenceNetwork  # noqa: E402
from pyprob.distributions import Normal, Uniform, Categorical, Mixture, Poisson, Bernoulli  # noqa: E402
from pyprob.nn import Batch  # noqa: E402

torch.set_num_threads(4)


class GaussianWithUnknownMean(Model):
    def __init__(self, prior_mean=1, prior_stddev=math.sqrt(5), likelihood_stddev=math.sqrt(2)):
        self.prior_mean = prior_mean
        self.prior_stddev = prior_stddev
        self.likelihood_stddev = likelihood_stddev
        super().__init__('Gaussian with unknown mean')

    def forward(self):
        mu = pyprob.sample(Normal(self.prior_mean, self.prior_stddev))
        likelihood = Normal(mu, self.likelihood_stddev)
        pyprob.observe(likelihood, name='obs0')
        pyprob.observe(likelihood, name='obs1')
        return mu


class GaussianWithUnknownMeanMarsaglia(Model):
    def __init__(self, prior_mean=1, prior_stddev=math.sqrt(5), likelihood_stddev=math.sqrt(2)):
        self.prior_mean = prior_mean
        self.prior_stddev = prior_stddev
        self.likelihood_stddev = likelihood_stddev
        super().__init__('Gaussian with unknown mean (Marsaglia)')

    def marsaglia(self, mean, stddev):
        uniform = Uniform(-1, 1)
        s = 1
        while float(s) >= 1:
            x = pyprob.sample(uniform)
            y = pyprob.sample(uniform)
            s = x * x + y * y
        return mean + stddev * (x * torch.sqrt(-2 * torch.log(s) / s))

    def forward(self):
        mu = self.marsaglia(self.prior_mean, self.prior_stddev)
        likelihood = Normal(mu, self.likelihood_stddev)
        pyprob.observe(likelihood, name='obs0')
        pyprob.observe(likelihood, name='obs1')
        return mu


class CategoricalThenNormal(Model):
    """c ~ Categorical(3); mu ~ Normal(c, 1.5); two Normal observations."""

    def __init__(self):
        super().__init__('Categorical then Normal')

    def forward(self):
        c = pyprob.sample(Categorical([0.2, 0.3, 0.5]))
        mu = pyprob.sample(Normal(c.float() * 2.0 - 1.0, 1.5))
        likelihood = Normal(mu, 0.8)
        pyprob.observe(likelihood, name='obs0')
        pyprob.observe(likelihood, name='obs1')
        return mu


class PoissonThenNormal(Model):
    """n ~ Poisson(4); mu ~ Normal(n / 2, 1); two Normal observations (ProposalPoissonTruncatedNormalMixture)."""

    def __init__(self):
        super().__init__('Poisson then Normal')

    def forward(self):
        n = pyprob.sample(Poisson(4.0))
        mu = pyprob.sample(Normal(n * 0.5, 1.0))
        likelihood = Normal(mu, 0.8)
        pyprob.observe(likelihood, name='obs0')
        pyprob.observe(likelihood, name='obs1')
        return mu


class BernoulliThenNormal(Model):
    """b ~ Bernoulli(0.3); mu ~ Normal(2 b - 1, 1); two Normal observations (ProposalBernoulliBernoulli)."""

    def __init__(self):
        super().__init__('Bernoulli then Normal')

    def forward(self):
        b = pyprob.sample(Bernoulli(0.3))
        mu = pyprob.sample(Normal(b * 2.0 - 1.0, 1.0))
        likelihood = Normal(mu, 0.8)
        pyprob.observe(likelihood, name='obs0')
        pyprob.observe(likelihood, name='obs1')
        return mu


def prior_params(dist):
    if isinstance(dist, Normal):
        return 'Normal', [float(dist.mean), float(dist.stddev)]
    if isinstance(dist, Uniform):
        return 'Uniform', [float(dist.low), float(dist.high)]
    if isinstance(dist, Categorical):
        return 'Categorical', [float(p) for p in dist.probs.view(-1)]
    if isinstance(dist, Poisson):
        return 'Poisson', [float(dist.rate)]
    if isinstance(dist, Bernoulli):
        return 'Bernoulli', [float(dist.probs)]
    raise RuntimeError(dist.name)


def dump_batch(traces, obs_names):
    """Plain-array view of a list of reference Trace objects (ragged, trace-major)."""
    addresses, dist_names = [], []
    trace_len, addr_idx, values, prior = [], [], [], []
    for tr in traces:
        trace_len.append(tr.length_controlled)
        for v in tr.variables_controlled:
            if v.address not in addresses:
                addresses.append(v.address)
                dist_names.append(v.distribution.name)
            addr_idx.append(addresses.index(v.address))
            values.append(float(v.value))
            name, pp = prior_params(v.distribution)
            prior.append(pp)
    width = max(len(p) for p in prior)
    prior_arr = np.zeros((len(prior), width), np.float32)
    for i, p in enumerate(prior):
        prior_arr[i, :len(p)] = p
    obs = np.array([[float(tr.named_variables[n].value) for n in obs_names] for tr in traces], np.float32)
    arrays = dict(trace_len=np.array(trace_len, np.int32), addr_idx=np.array(addr_idx, np.int32),
                  values=np.array(values, np.float32), prior=prior_arr, obs=obs)
    meta = dict(addresses=addresses, dist_names=dist_names, obs_names=list(obs_names))
    return arrays, meta


def run_case(case, model, lstm_dim, train_traces, train_batch, batch_size, num_particles, observe, network='lstm',
             lstm_depth=1, obs_emb=None):
    print('=' * 30, case)
    pyprob.seed(123)
    obs_emb = obs_emb or {'obs0': {'dim': 32}, 'obs1': {'dim': 32}}
    model.learn_inference_network(num_traces=train_traces, batch_size=train_batch,
                                  observe_embeddings=obs_emb,
                                  inference_network=(InferenceNetwork.LSTM if network == 'lstm' else
                                                     InferenceNetwork.FEEDFORWARD),
                                  lstm_dim=lstm_dim, lstm_depth=lstm_depth, proposal_mixture_components=10,
                                  learning_rate_init=1e-3, weight_decay=0.)
    net = model._inference_network
    net.train()

    # ---- one fresh minibatch, reference loss + grads -------------------------------------------------
    gen = model._trace_generator(trace_mode=pyprob.TraceMode.PRIOR_FOR_INFERENCE_NETWORK)
    traces = [next(gen) for _ in range(batch_size)]
    batch = Batch(traces)
    net._polymorph(batch)  # new addresses in this batch get layers (as optimize() would do, inference_network.py:479)

    sd = {k: v.detach().cpu().numpy().copy() for k, v in net.state_dict().items()}

    rec = {'lstm_in': [], 'lstm_out': [], 'log_prob': []}
    def lstm_hook(m, i, o):
        rec['lstm_in'].append(i[0].detach().numpy().copy())
        rec['lstm_out'].append(o[0].detach().numpy().copy())

    hook = net._layers_lstm.register_forward_hook(lstm_hook) if network == 'lstm' else None
    orig_mix_lp = Mixture.log_prob
    orig_cat_lp = Categorical.log_prob

    def mix_lp(self, value, sum=False):
        lp = orig_mix_lp(self, value, sum=sum)
        rec['log_prob'].append(lp.detach().numpy().copy().reshape(-1))
        return lp

    def cat_lp(self, value, sum=False):
        lp = orig_cat_lp(self, value, sum=sum)
        rec['log_prob'].append(lp.detach().numpy().copy().reshape(-1))
        return lp

    orig_ber_lp = Bernoulli.log_prob

    def ber_lp(self, value, sum=False):
        lp = orig_ber_lp(self, value, sum=sum)
        if self.probs.dim() == 2:      # the proposal (probs [B, 1]); the broadcast [B, B] result of _loss is kept as is
            rec['log_prob'].append(lp.detach().numpy().copy().reshape(-1))
        return lp

    Mixture.log_prob = mix_lp
    Categorical.log_prob = cat_lp
    Bernoulli.log_prob = ber_lp
    net.zero_grad()
    ok, loss = net._loss(batch)
    assert ok
    loss.backward()
    Mixture.log_prob = orig_mix_lp
    Categorical.log_prob = orig_cat_lp
    Bernoulli.log_prob = orig_ber_lp
    if hook is not None:
        hook.remove()
    names = [n for n, _ in net.named_parameters()]
    grads = {}
    for n, p in net.named_parameters():
        grads[n] = (p.grad.detach().numpy().copy() if p.grad is not None else np.zeros(p.shape, np.float32))
    has_grad = [int(p.grad is not None) for _, p in net.named_parameters()]

    arrays, meta = dump_batch(traces, list(obs_emb.keys()))
    # sub-batch structure in reference order (pyprob/nn/dataset.py:32-36): list of lists of trace indices
    index_of = {id(t): i for i, t in enumerate(traces)}
    sub_batches = [[index_of[id(t)] for t in sb] for sb in batch.sub_batches]
    meta['sub_batches'] = sub_batches
    meta['param_names'] = names
    meta['has_grad'] = has_grad
    meta['lstm_dim'] = lstm_dim if network == 'lstm' else 0
    meta['lstm_depth'] = lstm_depth
    meta['network'] = network
    meta['mixture_components'] = 10
    meta['observe_embedding_dims'] = {k: v['dim'] for k, v in obs_emb.items()}
    meta['observe_embedding_depths'] = {k: v.get('depth', 2) for k, v in obs_emb.items()}
    meta['num_params'] = int(sum(p.numel() for p in net.parameters()))
    meta['python'] = sys.version.split()[0]
    meta['torch'] = torch.__version__
    meta['pyprob'] = pyprob.__version__

    np.savez_compressed(os.path.join(HERE, case + '_net.npz'), **{'p%d' % i: sd[n] for i, n in enumerate(sd.keys())})
    meta['state_dict_names'] = list(sd.keys())
    np.savez_compressed(os.path.join(HERE, case + '_batch.npz'), **arrays)
    loss_arrays = {'loss': np.array(float(loss), np.float64)}
    for i, (xi, xo) in enumerate(zip(rec['lstm_in'], rec['lstm_out'])):
        loss_arrays['lstm_in_%d' % i] = xi
        loss_arrays['lstm_out_%d' % i] = xo
    # log_prob records arrive in order (sub_batch, time_step); flatten with an index
    lp_index, k = [], 0
    for si, sb in enumerate(batch.sub_batches):
        for t in range(sb[0].length_controlled):
            loss_arrays['lp_%d_%d' % (si, t)] = rec['log_prob'][k]
            lp_index.append([si, t])
            k += 1
    assert k == len(rec['log_prob'])
    for i, n in enumerate(names):
        loss_arrays['g%d' % i] = grads[n]
    np.savez_compressed(os.path.join(HERE, case + '_loss.npz'), **loss_arrays)
    print(case, 'loss', float(loss), 'params', meta['num_params'], 'sub-batches', len(sub_batches))

    # ---- importance sampling with the inference network ---------------------------------------------
    net.eval()
    steps = []
    orig_infer_step = net._infer_step

    def infer_step(variable, prev_variable=None, proposal_min_train_iterations=None):
        d = orig_infer_step(variable, prev_variable=prev_variable, proposal_min_train_iterations=proposal_min_train_iterations)
        steps.append((variable.address, d))
        return d

    net._infer_step = infer_step
    pyprob.seed(7)
    is_rows = dict(trace_len=[], addr=[], value=[], prior=[], prior_lp=[], prop_lp=[], prop_params=[], lw=[], obs_lw=[], result=[])
    gen = model._trace_generator(trace_mode=pyprob.TraceMode.POSTERIOR,
                                 inference_engine=InferenceEngine.IMPORTANCE_SAMPLING_WITH_INFERENCE_NETWORK,
                                 inference_network=net, observe=observe)
    is_addresses = []
    with torch.no_grad():
        for _ in range(num_particles):
            steps.clear()
            tr = next(gen)
            assert len(steps) == tr.length_controlled
            is_rows['trace_len'].append(tr.length_controlled)
            is_rows['lw'].append(float(tr.log_importance_weight))
            is_rows['obs_lw'].append(sum(float(v.log_importance_weight) for v in tr.variables_observed))
            is_rows['result'].append(float(tr.result))
            for v, (addr, d) in zip(tr.variables_controlled, steps):
                assert addr == v.address
                if addr not in is_addresses:
                    is_addresses.append(addr)
                is_rows['addr'].append(is_addresses.index(addr))
                is_rows['value'].append(float(v.value))
                _, pp = prior_params(v.distribution)
                is_rows['prior'].append(pp + [0.0] * (3 - len(pp)))
                is_rows['prior_lp'].append(float(v.log_prob))
                is_rows['prop_lp'].append(float(d.log_prob(v.value, sum=True)))
                if isinstance(d, Mixture):
                    comps = d.distributions
                    if hasattr(comps[0], 'mean_non_truncated'):
                        mus = [float(c.mean_non_truncated.view(-1)[0]) for c in comps]
                        sds = [float(c.stddev_non_truncated.view(-1)[0]) for c in comps]
                    else:
                        mus = [float(c.mean.view(-1)[0]) for c in comps]
                        sds = [float(c.stddev.view(-1)[0]) for c in comps]
                    pr = [float(p) for p in d.probs.view(-1)]
                    is_rows['prop_params'].append(mus + sds + pr)
                else:
                    pr = [float(p) for p in d.probs.view(-1)]
                    is_rows['prop_params'].append(pr + [0.0] * (30 - len(pr)))
    net._infer_step = orig_infer_step
    is_arrays = dict(trace_len=np.array(is_rows['trace_len'], np.int32), addr=np.array(is_rows['addr'], np.int32),
                     value=np.array(is_rows['value'], np.float32), prior=np.array(is_rows['prior'], np.float32),
                     prior_lp=np.array(is_rows['prior_lp'], np.float64), prop_lp=np.array(is_rows['prop_lp'], np.float64),
                     prop_params=np.array(is_rows['prop_params'], np.float32), lw=np.array(is_rows['lw'], np.float64),
                     obs_lw=np.array(is_rows['obs_lw'], np.float64), result=np.array(is_rows['result'], np.float32),
                     observe=np.array([float(observe[n]) for n in obs_emb.keys()], np.float32))
    np.savez_compressed(os.path.join(HERE, case + '_is.npz'), **is_arrays)
    meta['is_addresses'] = is_addresses
    meta['lp_index'] = lp_index
    with open(os.path.join(HERE, case + '_meta.json'), 'w') as f:
        json.dump(meta, f, indent=1)
    print(case, 'IS particles', num_particles, 'mean lw', float(np.mean(is_rows['lw'])))


if __name__ == '__main__':
    obs = {'obs0': 8, 'obs1': 9}
    only = sys.argv[1] if len(sys.argv) > 1 else None        # e.g. `poi` / `ff`: the other fixtures stay byte-identical
    if only == 'ber':
        # ProposalBernoulliBernoulli: in _loss its probs [B, 1] broadcast against values [B] to a [B, B] log_prob matrix
        # (every proposal scored against every value of the sub-batch step); recorded as the reference computes it.
        torch.distributions.Distribution.set_default_validate_args(False)
        run_case('ber', BernoulliThenNormal(), 64, 1280, 64, 48, 32, {'obs0': 1.2, 'obs1': 0.7})
        sys.exit(0)
    if only == 'gumd':
        # observe embeddings of depth 3 and 1 (EmbeddingFeedForward(num_layers=depth), inference_network.py:110-118)
        run_case('gumd', GaussianWithUnknownMean(), 32, 1280, 64, 48, 24, obs,
                 obs_emb={'obs0': {'dim': 32, 'depth': 3}, 'obs1': {'dim': 16, 'depth': 1}})
        sys.exit(0)
    if only == 'gumm2':
        # nn.LSTM(I, H, 2) (learn_inference_network(lstm_depth=2), inference_network_lstm.py:31): stacked layers
        run_case('gumm2', GaussianWithUnknownMeanMarsaglia(), 32, 1280, 64, 48, 24, obs, lstm_depth=2)
        sys.exit(0)
    if only == 'ff':
        # InferenceNetworkFeedForward (pyprob/nn/inference_network_feedforward.py): heads read the observe embedding
        run_case('ff', GaussianWithUnknownMeanMarsaglia(), 0, 2560, 128, 96, 48, obs, network='feedforward')
        run_case('ffc', CategoricalThenNormal(), 0, 1280, 64, 48, 32, {'obs0': 1.2, 'obs1': 0.7}, network='feedforward')
        sys.exit(0)
    if only is None:
        run_case('gum', GaussianWithUnknownMean(), 64, 1280, 64, 64, 64, obs)
        run_case('gumm', GaussianWithUnknownMeanMarsaglia(), 64, 2560, 128, 96, 48, obs)
        run_case('cat', CategoricalThenNormal(), 64, 1280, 64, 48, 32, {'obs0': 1.2, 'obs1': 0.7})
    # The proposal of a Poisson variable is a continuous TruncatedNormal mixture; Poisson.log_prob of its draw is what
    # the reference computes (state.py:211), but torch >= 1.8 rejects non-integer values unless argument validation is
    # off (the reference predates that check).
    torch.distributions.Distribution.set_default_validate_args(False)
    run_case('poi', PoissonThenNormal(), 64, 1280, 64, 48, 32, {'obs0': 2.2, 'obs1': 1.7})
