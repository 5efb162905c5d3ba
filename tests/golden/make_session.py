#!/usr/bin/env python3
"""Record TRAINING SESSIONS of the stock reference (pyprob v1.5.0, CPU) for the device twin of the binding tests.

    python tests/golden/make_session.py          (build container only: needs /root/reference)

The reference cannot travel to the GPU box in any form, so what `pyprob.Model.learn_inference_network` does with its OWN
`InferenceNetworkLSTM` is recorded here as data and replayed on the MI355X through `pyprob_amd.hip_network._HipNetworkMixin`
- the class body `pyprob_amd/binding.py` puts under the real pyprob - by tests/test_gpu_binding_session.py (and, with the
oracle-backed CPU kernels, by tests/test_binding_session.py in this container). Nothing of the reference is copied: its
public API is called and inputs / outputs are written down.

Per program (gum: tests/test_inference.py:97-109, gumm: :252-275 of the reference) `session_<case>.npz` + `session_<case>.json`:
  * the parameters every `_init_layers*` / `_polymorph` call CREATED (names in creation order + initial values), keyed by the
    iteration that created them (-1: before the first minibatch) - the stand-in module tree grows with the same values;
  * every minibatch the reference's DataLoader produced (`Batch.traces` in plain arrays: lengths, address index, values,
    prior parameters, observed values) and the loss `_loss(batch)` returned for it (inference_network_lstm.py:136-220);
  * after `optimize` (Adam, lr 1e-3, inference_network.py:381-599): `named_parameters()` order, the final `state_dict`,
    `_history_train_loss`, per-address `_total_train_iterations`, the optimizer's `exp_avg` of two parameters;
  * the stock network's `_infer_step` proposals for recorded particles: per controlled variable the proposal log-prob of the
    value the reference drew, and the traces' log importance weights (state.py:203-219, trace.py:123-125).
"""
import json
import math
import os
import sys
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(REPO, 'oracle', 'refstubs'))
sys.path.insert(1, '/root/reference')

import numpy as np  # noqa: E402
import torch  # noqa: E402
import pyprob  # noqa: E402
from pyprob import InferenceEngine, InferenceNetwork, Model  # noqa: E402
from pyprob.distributions import Categorical, Normal, Uniform  # noqa: E402
from pyprob.nn import InferenceNetworkFeedForward, InferenceNetworkLSTM  # noqa: E402

torch.set_num_threads(2)
LSTM_DIM, BATCH, ITERATIONS, PARTICLES = 32, 32, 12, 48
EMB = {'obs0': {'dim': 16}, 'obs1': {'dim': 16}}
OBSERVE = {'gum': {'obs0': 8.0, 'obs1': 9.0}, 'gumm': {'obs0': 4.0, 'obs1': 5.0}, 'ffcat': {'obs0': 1.0, 'obs1': 1.5},
           'gumm2': {'obs0': 6.0, 'obs1': 7.0}}


class GaussianWithUnknownMean(Model):
    def __init__(self):
        super().__init__('Gaussian with unknown mean')

    def forward(self):
        mu = pyprob.sample(Normal(1, math.sqrt(5)))
        likelihood = Normal(mu, math.sqrt(2))
        pyprob.observe(likelihood, name='obs0')
        pyprob.observe(likelihood, name='obs1')
        return mu


class GaussianWithUnknownMeanMarsaglia(Model):
    def __init__(self):
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
        mu = self.marsaglia(1, math.sqrt(5))
        likelihood = Normal(mu, math.sqrt(2))
        pyprob.observe(likelihood, name='obs0')
        pyprob.observe(likelihood, name='obs1')
        return mu


class CategoricalThenNormal(Model):      # the program of the `cat` golden case
    def __init__(self):
        super().__init__('categorical then normal')

    def forward(self):
        c = pyprob.sample(Categorical([0.2, 0.3, 0.5]))
        mu = pyprob.sample(Normal(c.float() * 2.0 - 1.0, 1.5))
        likelihood = Normal(mu, 0.8)
        pyprob.observe(likelihood, name='obs0')
        pyprob.observe(likelihood, name='obs1')
        return mu


def dist_params(d):
    if d.name == 'Normal':
        return [float(d.mean), float(d.stddev), 0.0]
    if d.name == 'Uniform':
        return [float(d.low), float(d.high), 0.0]
    if d.name == 'Categorical':
        return [float(p) for p in d._probs.reshape(-1)]          # (three categories in the recorded program)
    raise ValueError(d.name)


def record(case, program, seed, network='lstm', lstm_depth=1):
    rec = dict(created={}, batches=[], losses=[], addresses=[], dist_names=[])
    arrays = {}
    state = dict(iteration=-1, known=[])

    def note_created(net):
        names = [n for n, _ in net.named_parameters()]
        new = [n for n in names if n not in state['known']]
        if new:
            sd = dict(net.named_parameters())
            key = str(state['iteration'])
            rec['created'].setdefault(key, [])
            for n in new:
                arrays['init_%d' % len(state['known'])] = sd[n].detach().cpu().numpy().copy()
                rec['created'][key].append(n)
                state['known'].append(n)

    base = InferenceNetworkLSTM if network == 'lstm' else InferenceNetworkFeedForward

    class Recording(base):
        def _init_layers(self):
            super()._init_layers()
            note_created(self)

        def _polymorph(self, batch):
            state['iteration'] += 1
            changed = super()._polymorph(batch)
            note_created(self)
            return changed

        def _loss(self, batch):
            i = state['iteration']
            trace_len, addr_idx, values, prior, obs = [], [], [], [], []
            for tr in batch.traces:
                trace_len.append(tr.length_controlled)
                for v in tr.variables_controlled:
                    if v.address not in rec['addresses']:
                        rec['addresses'].append(v.address)
                        rec['dist_names'].append(v.distribution.name)
                    addr_idx.append(rec['addresses'].index(v.address))
                    values.append(float(v.value))
                    prior.append(dist_params(v.distribution))
                obs.append([float(tr.named_variables[n].value) for n in EMB])
            arrays['b%d_trace_len' % i] = np.asarray(trace_len, np.int32)
            arrays['b%d_addr_idx' % i] = np.asarray(addr_idx, np.int32)
            arrays['b%d_values' % i] = np.asarray(values, np.float32)
            arrays['b%d_prior' % i] = np.asarray(prior, np.float32)
            arrays['b%d_obs' % i] = np.asarray(obs, np.float32)
            ok, loss = super()._loss(batch)
            assert ok
            rec['losses'].append(float(loss))
            return ok, loss

    import pyprob.model as M
    slot = 'InferenceNetworkLSTM' if network == 'lstm' else 'InferenceNetworkFeedForward'
    stock = getattr(M, slot)
    setattr(M, slot, Recording)
    try:
        pyprob.seed(seed)
        model = program()
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            model.learn_inference_network(num_traces=ITERATIONS * BATCH, batch_size=BATCH, observe_embeddings=EMB,
                                          inference_network=InferenceNetwork.LSTM if network == 'lstm' else InferenceNetwork.FEEDFORWARD,
                                          lstm_dim=LSTM_DIM, lstm_depth=lstm_depth, learning_rate_init=1e-3)
    finally:
        setattr(M, slot, stock)
    net = model._inference_network
    assert len(rec['losses']) == ITERATIONS and np.allclose(rec['losses'], net._history_train_loss)
    names = [n for n, _ in net.named_parameters()]
    for k, (n, p) in enumerate(net.named_parameters()):
        arrays['final_%d' % k] = p.detach().cpu().numpy().copy()
    ost = net._optimizer.state
    watch = [names.index('_layers_lstm.weight_ih_l0') if network == 'lstm' else 0, len(names) - 1]
    for k in watch:
        p = dict(net.named_parameters())[names[k]]
        arrays['exp_avg_%d' % k] = ost[p]['exp_avg'].cpu().numpy().copy()
        arrays['exp_avg_sq_%d' % k] = ost[p]['exp_avg_sq'].cpu().numpy().copy()
    # ---- importance sampling with the trained stock network: what _infer_step proposed for the values the reference drew ------
    observe = OBSERVE[case]
    pyprob.seed(seed + 1)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        post = model.posterior(PARTICLES, inference_engine=InferenceEngine.IMPORTANCE_SAMPLING_WITH_INFERENCE_NETWORK, observe=observe)
    is_len, is_addr, is_val, is_prior, is_logq, is_lw, is_result = [], [], [], [], [], [], []
    for k in range(post.length):
        tr = post._get_value(k)
        is_len.append(tr.length_controlled)
        is_lw.append(float(tr.log_importance_weight))
        is_result.append(float(tr.result))
        for v in tr.variables_controlled:
            assert v.address in rec['addresses'], 'a particle visited an address training never saw: pick another seed'
            is_addr.append(rec['addresses'].index(v.address))
            is_val.append(float(v.value))
            is_prior.append(dist_params(v.distribution))
            # state.py:211-217: log_importance_weight of the variable = log p(v) - log q(v)
            is_logq.append(float(v.log_prob) - float(v.log_importance_weight))
    arrays.update(is_trace_len=np.asarray(is_len, np.int32), is_addr_idx=np.asarray(is_addr, np.int32),
                  is_values=np.asarray(is_val, np.float64), is_prior=np.asarray(is_prior, np.float64),
                  is_logq=np.asarray(is_logq, np.float64), is_lw=np.asarray(is_lw, np.float64), is_result=np.asarray(is_result, np.float64))
    meta = dict(case=case, network=network, lstm_dim=LSTM_DIM, lstm_depth=lstm_depth, mixture_components=10, batch_size=BATCH, iterations=ITERATIONS,
                observe_embeddings=EMB, obs_names=list(EMB), observe=observe, learning_rate=1e-3, weight_decay=0.0, optimizer='ADAM',
                created=rec['created'], losses=rec['losses'], addresses=rec['addresses'], dist_names=rec['dist_names'],
                param_order=names, exp_avg_watch=watch, history_num_params=net._history_num_params,
                total_train_iterations={a: int(l._total_train_iterations) for a, l in net._layers_proposal.items()},
                total_train_traces=int(net._total_train_traces), python=sys.version.split()[0], torch=torch.__version__,
                pyprob=pyprob.__version__, seed=seed)
    np.savez_compressed(os.path.join(HERE, 'session_%s.npz' % case), **arrays)
    with open(os.path.join(HERE, 'session_%s.json' % case), 'w') as f:
        json.dump(meta, f, indent=1)
    print(case, 'losses', ['%.4f' % l for l in rec['losses']], 'params', len(names), 'created at', sorted(rec['created'], key=int))


if __name__ == '__main__':
    only = sys.argv[1:]
    for case, program, seed, network, depth in (('gum', GaussianWithUnknownMean, 5, 'lstm', 1), ('gumm', GaussianWithUnknownMeanMarsaglia, 7, 'lstm', 1),
                                                ('ffcat', CategoricalThenNormal, 9, 'feedforward', 1),
                                                ('gumm2', GaussianWithUnknownMeanMarsaglia, 11, 'lstm', 2)):      # nn.LSTM(I, 32, 2)
        if not only or case in only:
            record(case, program, seed, network, depth)
