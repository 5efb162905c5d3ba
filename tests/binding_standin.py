"""TEST INFRASTRUCTURE: the data-holder side of the binding's device twin.

The reference (pyprob, Python) cannot exist on the GPU box, so the class body that ships under the real pyprob -
`pyprob_amd.hip_network._HipNetworkMixin` - is composed here with a module tree that holds DATA only:
`RecordedInferenceNetworkLSTM` has the attribute names, layer class names, registration order and `state_dict` names of
`pyprob.nn.InferenceNetworkLSTM` (pyprob/nn/inference_network.py:25-80, inference_network_lstm.py:11-80) and creates every
parameter with the INITIAL VALUE the stock reference gave it in a recorded session (tests/golden/make_session.py). It has
no forward arithmetic: losses, gradients, optimizer steps and proposals come from the mixin (the HIP engine) and are
compared with what the stock reference computed in that session.
`SessionBatch` rebuilds the recorded `Batch.traces` as duck-typed trace records (pyprob/nn/dataset.py:21-37)."""
import json
import os

import numpy as np
import torch
import torch.nn as nn

from pyprob_amd import distributions as D
from pyprob_amd.hip_network import _HipNetworkMixin
from pyprob_amd.nn import Batch
from pyprob_amd.trace import Trace, Variable

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_session(case):
    with open(os.path.join(GOLDEN, 'session_%s.json' % case)) as f:
        meta = json.load(f)
    arrays = dict(np.load(os.path.join(GOLDEN, 'session_%s.npz' % case)))
    init, k = {}, 0
    for it in sorted(meta['created'], key=int):
        for name in meta['created'][it]:
            init[name] = arrays['init_%d' % k]
            k += 1
    final = {n: arrays['final_%d' % i] for i, n in enumerate(meta['param_order'])}
    return meta, arrays, init, final


# ---- layer classes: the NAMES are the reference's (the mixin dispatches on type(layer).__name__) ---------------------------
class EmbeddingFeedForward(nn.Module):
    """pyprob/nn/embedding_feedforward.py:8-33 as a parameter holder: `_layers` = nn.Linear list, `_input_dim`, `_output_dim`."""

    def __init__(self, weights_and_biases):
        super().__init__()
        layers = []
        for w, b in weights_and_biases:
            lin = nn.Linear(w.shape[1], w.shape[0])
            with torch.no_grad():
                lin.weight.copy_(torch.as_tensor(w))
                lin.bias.copy_(torch.as_tensor(b))
            layers.append(lin)
        self._input_dim = int(weights_and_biases[0][0].shape[1])
        self._output_dim = int(weights_and_biases[-1][0].shape[0])
        self._layers = nn.ModuleList(layers)


class _Proposal(nn.Module):
    def __init__(self, weights_and_biases):
        super().__init__()
        self._ff = EmbeddingFeedForward(weights_and_biases)
        self._total_train_iterations = 0          # proposal_normal_normal_mixture.py:17


class ProposalNormalNormalMixture(_Proposal):
    pass


class ProposalUniformTruncatedNormalMixture(_Proposal):
    pass


class ProposalCategoricalCategorical(_Proposal):
    pass


_PROPOSAL_CLASS = {'Normal': ProposalNormalNormalMixture, 'Uniform': ProposalUniformTruncatedNormalMixture,
                   'Categorical': ProposalCategoricalCategorical}


def _linears(init, prefix):
    out, k = [], 0
    while prefix + '_layers.%d.weight' % k in init:
        out.append((init[prefix + '_layers.%d.weight' % k], init[prefix + '_layers.%d.bias' % k]))
        k += 1
    return out


class RecordedInferenceNetworkLSTM(nn.Module):
    """Attribute for attribute what `InferenceNetwork.__init__` + `InferenceNetworkLSTM.__init__` set up, in their order
    (it decides the order of `named_parameters()`, which `_distributed_sync_grad` and the optimizer state rely on)."""

    def __init__(self, session, lstm_dim=512, lstm_depth=1, sample_embedding_dim=4, address_embedding_dim=64,
                 distribution_type_embedding_dim=8, proposal_mixture_components=10):
        super().__init__()
        self._session_meta, self._session_init = session
        self._layers_observe_embedding = nn.ModuleDict()
        self._layers_observe_embedding_final = None
        self._layers_pre_generated = False
        self._layers_initialized = False
        self._observe_embedding_dim = None
        self._optimizer = None
        self._optimizer_type = None
        self._momentum = None
        self._weight_decay = None
        self._learning_rate_init = None
        self._total_train_traces = 0
        self._total_train_iterations = 0
        self._history_num_params = []
        self._history_num_params_trace = []
        self._distributed_train_loss = torch.tensor(0.)
        self._distributed_history_train_loss = []
        self._distributed_history_train_loss_trace = []
        self._on_cuda = False
        self._device = torch.device('cpu')
        self._layers_proposal = nn.ModuleDict()
        self._layers_sample_embedding = nn.ModuleDict()
        self._layers_address_embedding = nn.ParameterDict()
        self._layers_distribution_type_embedding = nn.ParameterDict()
        self._layers_lstm = None
        self._lstm_dim = lstm_dim
        self._lstm_depth = lstm_depth
        self._sample_embedding_dim = sample_embedding_dim
        self._address_embedding_dim = address_embedding_dim
        self._distribution_type_embedding_dim = distribution_type_embedding_dim
        self._proposal_mixture_components = proposal_mixture_components

    def _init_layers_observe_embedding(self, observe_embeddings, example_trace=None):      # inference_network.py:80-130
        init = self._session_init
        total = 0
        for name in observe_embeddings:
            layer = EmbeddingFeedForward(_linears(init, '_layers_observe_embedding.%s.' % name))
            self._layers_observe_embedding[name] = layer
            total += layer._output_dim
        self._observe_embedding_dim = total
        self._layers_observe_embedding_final = EmbeddingFeedForward(_linears(init, '_layers_observe_embedding_final.'))

    def _init_layers(self):                                                                 # inference_network_lstm.py:29-32
        init = self._session_init
        in_dim = self._observe_embedding_dim + self._sample_embedding_dim + 2 * (self._address_embedding_dim + self._distribution_type_embedding_dim)
        self._layers_lstm = nn.LSTM(in_dim, self._lstm_dim, self._lstm_depth)
        with torch.no_grad():
            for n, p in self._layers_lstm.named_parameters():
                p.copy_(torch.as_tensor(init['_layers_lstm.' + n]))

    def _polymorph(self, batch):                                                            # inference_network_lstm.py:34-80
        init = self._session_init
        layers_changed = False
        for sub_batch in batch.sub_batches:
            for variable in sub_batch[0].variables_controlled:
                address, dname = variable.address, variable.distribution.name
                if address not in self._layers_address_embedding:
                    self._layers_address_embedding[address] = nn.Parameter(torch.as_tensor(init['_layers_address_embedding.' + address]).clone())
                if dname not in self._layers_distribution_type_embedding:
                    self._layers_distribution_type_embedding[dname] = nn.Parameter(
                        torch.as_tensor(init['_layers_distribution_type_embedding.' + dname]).clone())
                if address not in self._layers_proposal:
                    self._layers_sample_embedding[address] = EmbeddingFeedForward(_linears(init, '_layers_sample_embedding.%s.' % address))
                    self._layers_proposal[address] = _PROPOSAL_CLASS[dname](_linears(init, '_layers_proposal.%s._ff.' % address))
                    layers_changed = True
        if layers_changed:
            self._history_num_params.append(sum(p.numel() for p in self.parameters()))
            self._history_num_params_trace.append(self._total_train_traces)
        return layers_changed


class RecordedInferenceNetworkFeedForward(nn.Module):
    """`InferenceNetwork.__init__` + `InferenceNetworkFeedForward.__init__` (inference_network_feedforward.py:11-19): observe
    embeddings and one proposal layer per address, nothing else."""

    def __init__(self, session, proposal_mixture_components=10, **unused):
        super().__init__()
        self._session_meta, self._session_init = session
        self._layers_observe_embedding = nn.ModuleDict()
        self._layers_observe_embedding_final = None
        self._layers_pre_generated = False
        self._layers_initialized = False
        self._observe_embedding_dim = None
        self._optimizer = None
        self._optimizer_type = None
        self._momentum = None
        self._weight_decay = None
        self._learning_rate_init = None
        self._total_train_traces = 0
        self._total_train_iterations = 0
        self._history_num_params = []
        self._history_num_params_trace = []
        self._distributed_train_loss = torch.tensor(0.)
        self._distributed_history_train_loss = []
        self._distributed_history_train_loss_trace = []
        self._on_cuda = False
        self._device = torch.device('cpu')
        self._layers_proposal = nn.ModuleDict()
        self._proposal_mixture_components = proposal_mixture_components

    _init_layers_observe_embedding = RecordedInferenceNetworkLSTM._init_layers_observe_embedding

    def _init_layers(self):
        pass

    def _polymorph(self, batch):                                                            # inference_network_feedforward.py:21-50
        init = self._session_init
        layers_changed = False
        for sub_batch in batch.sub_batches:
            for variable in sub_batch[0].variables_controlled:
                address, dname = variable.address, variable.distribution.name
                if address not in self._layers_proposal:
                    self._layers_proposal[address] = _PROPOSAL_CLASS[dname](_linears(init, '_layers_proposal.%s._ff.' % address))
                    layers_changed = True
        if layers_changed:
            self._history_num_params.append(sum(p.numel() for p in self.parameters()))
            self._history_num_params_trace.append(self._total_train_traces)
        return layers_changed


class RecordedLSTMHip(_HipNetworkMixin, RecordedInferenceNetworkLSTM):
    """= `binding.InferenceNetworkLSTMHip` with the recorded module tree in the place of `pyprob.nn.InferenceNetworkLSTM`."""
    _hip_kind = 'lstm'


class RecordedFeedForwardHip(_HipNetworkMixin, RecordedInferenceNetworkFeedForward):
    """= `binding.InferenceNetworkFeedForwardHip` over the recorded module tree."""
    _hip_kind = 'feedforward'


# ---- recorded minibatches ------------------------------------------------------------------------------------------------
def _distribution(name, p):
    if name == 'Normal':
        return D.Normal(float(p[0]), float(p[1]))
    if name == 'Uniform':
        return D.Uniform(float(p[0]), float(p[1]))
    if name == 'Categorical':
        return D.Categorical([float(x) for x in p])
    raise ValueError(name)


def traces_from_arrays(meta, trace_len, addr_idx, values, prior, obs):
    traces, r = [], 0
    for b, n in enumerate(trace_len):
        tr = Trace()
        for _ in range(int(n)):
            a = int(addr_idx[r])
            tr.add(Variable(distribution=_distribution(meta['dist_names'][a], prior[r]), value=float(values[r]),
                            address_base=meta['addresses'][a], address=meta['addresses'][a], control=True))
            r += 1
        for j, name in enumerate(meta['obs_names']):
            tr.add(Variable(value=float(obs[b, j]), address_base='obs_' + name, address='obs_' + name, name=name, observed=True))
        tr.end(None, 0.0)
        traces.append(tr)
    return traces


def session_batch(meta, arrays, i):
    """Iteration i's minibatch as the reference's DataLoader delivered it: Batch(traces) (sub-batched by address sequence)."""
    return Batch(traces_from_arrays(meta, arrays['b%d_trace_len' % i], arrays['b%d_addr_idx' % i], arrays['b%d_values' % i],
                                    arrays['b%d_prior' % i], arrays['b%d_obs' % i]))


def new_network(case, device, engine_factory=None):
    meta, arrays, init, final = load_session(case)
    base = RecordedLSTMHip if meta.get('network', 'lstm') == 'lstm' else RecordedFeedForwardHip
    cls = base
    if engine_factory is not None or device != cls._hip_device:
        name = base.__name__ + '_' + str(device).replace(':', '_')
        cls = type(name, (base,), dict(_hip_device=device, __module__=__name__))
        if engine_factory is not None:
            cls._hip_engine_factory = staticmethod(engine_factory)
        globals()[name] = cls               # (importable by name: torch.save pickles the class by reference)
    net = cls((meta, init), lstm_dim=meta['lstm_dim'], lstm_depth=meta.get('lstm_depth', 1),
              proposal_mixture_components=meta['mixture_components'])
    net._init_layers_observe_embedding(meta['observe_embeddings'])
    net._init_layers()
    net._layers_initialized = True
    return net, meta, arrays, init, final


def replay_training(net, meta, arrays):
    """The loop body of pyprob's optimize() (inference_network.py:461-499) over the recorded minibatches: _polymorph ->
    (new) optimizer -> zero_grad -> _loss -> backward -> step. Returns the losses."""
    net._optimizer_type = meta['optimizer']
    net._learning_rate_init = meta['learning_rate']
    net._weight_decay = meta['weight_decay']
    net._momentum = 0.9
    losses = []
    for i in range(meta['iterations']):
        batch = session_batch(meta, arrays, i)
        layers_changed = net._polymorph(batch)
        if net._optimizer is None or layers_changed:
            net._create_optimizer()
        net._optimizer.zero_grad()
        success, loss = net._loss(batch)
        assert success
        loss.backward()
        net._optimizer.step()
        losses.append(float(loss.detach()))
        net._total_train_iterations += 1
        net._total_train_traces += batch.size
    return losses
